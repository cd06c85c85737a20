"""CPU, world_size 2 over gloo: the multi-GPU path of SURVEY.md section 8(e).  Frames are independent, so the path shards with no
collective: each rank packs its share of a batch (here through the emulator build of the SAME kernels), the host merges in
original index order -- running offsets from 12, first-wins dedup across ranks.  The merged archive body must be byte-identical
to what one handle writes for the same entries, and a duplicate whose copies land on different ranks is written once.
bench.py's aggregation (max time over ranks, sum of units) and the sharder itself are checked beside it."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "support"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _entries():
    import harness
    c = harness.Corpus()
    ents = [c.entry(500 + i, n, -1) for i, n in enumerate((70000, 3000, 140000, 65536, 0, 90001, 20000, 131072))]
    ents.append(ents[2])      # duplicates of entries that land on the OTHER rank (index 8 -> rank 0, its first copy 2 -> rank 0; 9 -> see below)
    ents.append(ents[1])      # index 9: copy of index 1
    ents.append(ents[0])      # index 10: copy of index 0
    return ents


def _worker(rank, world, port, emu_lib, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from zarc_amd import Engine, _lib, shard
    ents = _entries()
    shares = shard.assign([len(e) for e in ents], world)
    eng = Engine(0, emu_lib)
    eng.set_parameter(_lib.P_CHECKSUM_FLAG, 1)
    mine = eng.pack([ents[i] for i in shares[rank]])          # the hot path, on this rank's share only
    eng.close()
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)                       # host-side bookkeeping only: frames + digests to the merging rank
    t, u = bench.aggregate(float(rank + 1), 1000.0 * (1 << 20), dist, torch.device("cpu"))
    if rank == 0:
        body, records = shard.merge(shares, gathered)
        q.put((body, records, t, u))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_pack_over_ranks_equals_single_handle(emu_lib_path, oracle, world):
    from zarc_amd import Engine, _lib, shard
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, emu_lib_path, q)) for r in range(world)]
    for p in procs:
        p.start()
    body, records, t, u = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert t == float(world) and u == world * 1000.0 * (1 << 20)   # max over ranks, sum over ranks
    # the same entries through ONE handle, merged by the same host logic
    ents = _entries()
    eng = Engine(0, emu_lib_path)
    eng.set_parameter(_lib.P_CHECKSUM_FLAG, 1)
    single = eng.pack(ents)
    eng.close()
    body1, records1 = shard.merge([list(range(len(ents)))], [single])
    assert body == body1 and records == records1                 # byte-identical archive body, identical Frame records
    # duplicates: copies on different ranks (sizes differ -> greedy sharding) are written once, first index wins
    shares = shard.assign([len(e) for e in ents], world)
    owner = {i: d for d, s in enumerate(shares) for i in s}
    assert any(owner[a] != owner[b] for a, b in ((2, 8), (1, 9), (0, 10)))
    assert [r[3] for r in records[8:]] == [False, False, False] and all(r[3] for r in records[:8])
    assert records[8][:2] == records[2][:2] and records[9][:2] == records[1][:2] and records[10][:2] == records[0][:2]
    # offsets: running sums from 12 over what was written (content_frame.rs:22,45; encode.rs:65,75)
    pos = 12
    for (off, ln, dig, written), raw in zip(records, ents):
        assert dig == oracle.blake3(raw)
        if written:
            assert off == pos
            rc, out, used = oracle.zstd_decode(body[off - 12:off - 12 + ln], len(raw))
            assert rc == 0 and out == raw and used == ln
            pos += ln
    assert pos - 12 == len(body)


def _unpack_worker(rank, world, port, emu_lib, frames, raw_lens, digests, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from zarc_amd import Engine, shard
    shares = shard.assign_unpack(raw_lens, world)
    eng = Engine(0, emu_lib)
    mine = eng.unpack([frames[i] for i in shares[rank]], [raw_lens[i] for i in shares[rank]], [digests[i] for i in shares[rank]])
    eng.close()
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)                       # results to the caller's rank; the data path itself has no collective
    if rank == 0:
        q.put(shard.gather(shares, gathered))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_unpack_over_ranks_equals_single_handle(emu_lib_path, oracle, world):
    """The read side of section 8(e): the frames of an archive dealt to two ranks by uncompressed bytes, each rank decodes + verifies its
    share; merged results (bytes, digests, statuses) equal the single-handle ones -- with a corrupt frame and a wrong expected digest
    on rank 1's share (crates/zarc-cli/src/unpack.rs:62-88,118-120)."""
    from zarc_amd import Engine, _lib, shard
    ents = _entries()[:8]
    eng = Engine(0, emu_lib_path)
    eng.set_parameter(_lib.P_CHECKSUM_FLAG, 1)
    packed = eng.pack(ents)
    frames, digests, raw_lens = [p[0] for p in packed], [p[1] for p in packed], [len(e) for e in ents]
    shares = shard.assign_unpack(raw_lens, world)
    assert sorted(i for sh in shares for i in sh) == list(range(8)) and all(shares)
    big = [i for sh in shares[1:] for i in sh if raw_lens[i] > 50000]      # on ranks other than the one that gathers
    corrupt, wrong = big[0], big[-1]
    assert corrupt != wrong
    f = bytearray(frames[corrupt])
    f[len(f) // 2] ^= 0x40
    frames[corrupt] = bytes(f)
    digests[wrong] = bytes(32)
    single = eng.unpack(frames, raw_lens, digests)
    eng.close()
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_unpack_worker, args=(r, world, port, emu_lib_path, frames, raw_lens, digests, q)) for r in range(world)]
    for p in procs:
        p.start()
    merged = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [m[2] for m in merged] == [x[2] for x in single]                                   # statuses, in the caller's order
    assert [m[:2] for i, m in enumerate(merged) if i != corrupt] == [x[:2] for i, x in enumerate(single) if i != corrupt]
    for i, (out, dig, st) in enumerate(merged):
        if i == corrupt:
            assert st not in (_lib.FRAME_OK, _lib.FRAME_DIGEST)
        elif i == wrong:
            assert st == _lib.FRAME_DIGEST and out == ents[i] and dig == oracle.blake3(ents[i])
        else:
            assert st == _lib.FRAME_OK and out == ents[i] and dig == oracle.blake3(ents[i])


def test_sharder_covers_every_entry_once_and_balances():
    from zarc_amd import shard
    import random
    for g in (1, 2, 3, 8):
        assert shard.assign([1 << 20] * 10000, g) == [list(range(d, 10000, g)) for d in range(g)]      # equal sizes: index mod G
    rnd = random.Random(5)
    sizes = [int(65536 * 2 ** (rnd.random() * 8)) for _ in range(4000)]                                  # BASELINE configs[4] shape
    for g in (2, 8):
        shares = shard.assign(sizes, g)
        assert sorted(i for s in shares for i in s) == list(range(len(sizes)))
        loads = [sum(sizes[i] for i in s) for s in shares]
        assert max(loads) - min(loads) <= max(sizes) and all(s == sorted(s) for s in shares)


def test_bench_workload_gives_every_rank_the_same_mix():
    import argparse
    import bench
    a = argparse.Namespace(config="c2", entries=1000, size=1 << 20, kind=-1, gib=1.0)
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            sizes, cidx, kind, level, desc, n_global = bench.workload(a, r, world)
            assert len(sizes) == 1000 and n_global == 1000 * world and level == 3
            assert cidx == list(range(r * 1000, (r + 1) * 1000))       # contiguous corpus range: all four kinds in equal parts
            seen += cidx
        assert sorted(seen) == list(range(1000 * world))


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` without a launcher must start N ranks itself, as a child process (the driver's SCALE run may call
    it either way).  The command is torch.distributed.run on 127.0.0.1; on this GPU-less box the ranks then fail loudly at
    zarc_gpu_create (there is no CPU fallback) and the parent passes the failure on."""
    import subprocess
    import bench
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "2"], 29999)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "2"]
    assert cmd[cmd.index("--master-port") + 2] == os.path.join(ROOT, "bench.py")
    if torch.cuda.device_count() > 0:
        return                                                    # on a GPU box the real run is test_gpu_parity's business
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--entries", "4", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, timeout=600)
    err = r.stderr.decode(errors="replace")
    assert r.returncode != 0
    assert "WORLD_SIZE" not in err.split("Traceback")[0] or "launcher" not in err       # it did not stop at the old WORLD_SIZE check
    assert "no usable HIP device" in err or "nccl" in err.lower() or "cuda" in err.lower(), err[-2000:]
