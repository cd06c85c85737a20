#!/usr/bin/env python3
"""bench.py -- headline benchmark of the content engine (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU under torch.distributed.run.  Started that way (WORLD_SIZE set) this process IS a rank; started plainly
(`python bench.py --gpus 8`) it launches `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...
bench.py <same arguments>` as a CHILD process before torch or HIP are touched, relays the ranks' output (rank 0 prints the JSON
line) and exits with the child's code -- no exec from a process that has initialised the GPU.

metric  : uncompressed GiB/s of `zarc pack` at zstd level 3 (BLAKE3 + frame encode + XXH64), inputs resident in
          HBM; the same line carries the unpack rate, the compression ratio and the ratio vs libzstd -3.
workload: BASELINE.json configs[1]/[2] -- 10 000 x 1 MiB synthetic entries per GPU (kinds text / records / lz /
          random round-robin); `--entries` scales it down for quick runs.
          --config c4: configs[3] shape (4 MiB frames of the lz stream, level 9), --config c5: configs[4] shape (64 KiB .. 16 MiB
          log-uniform sizes, level 3) -- whatever count fits one GPU; their lines are kept under profiles/.
step    : one pack pass over the whole batch (timed region 1), and one unpack pass over its output (region 2).
scaling : weak -- the global entry list is dealt to the ranks by the same sharder the product uses (zarc::ShardedEncoder: index mod
          G for equal sizes, greedy by bytes otherwise; zarc_amd/shard.py), every rank packs its share; no collective on the data
          path; only the timing barrier / max-over-ranks uses torch.distributed (RCCL).
extras  : host_path (PCIe-inclusive rates through the host-pointer entry points, pageable and pinned), unpack_roofline,
          cpu_baseline (the reference's CPU path, C driver tests/support/cpu_baseline.c, 1 thread and all host cores).
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GIB = float(1 << 30)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def aggregate(local_seconds, local_units, dist=None, device=None):
    """Max time over ranks, sum of units over ranks (value = units / max time)."""
    if dist is None:
        return local_seconds, local_units
    import torch
    t = torch.tensor([local_seconds], dtype=torch.float64, device=device)
    u = torch.tensor([float(local_units)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def usable_cores():
    """Host threads this process can really run at once: the affinity mask, cut to the cgroup's CPU quota where there is one (the GPU boxes
    show 256 CPUs in the mask and grant a share of them: 256 threads then time-slice on that share, which is what round 3's all-cores
    figure measured)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                       # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())                          # cgroup v1
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(sample, size, first_index, level, kind=-1):
    """The reference's CPU path on this box's host cores (rank 0, N = 1 only), bounded sample: tests/support/cpu_baseline.c --
    one CCtx with session resets (content_frame.rs:37-41), decompressStream in 131 075 / 131 072-byte steps
    (zstd_iterator.rs:88-153), BLAKE3 on both sides.  libzstd = the fastest build found on the box (they differ 3x)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "support"))
    import harness
    zs = harness.libzstds()
    if not zs:
        return None, None
    probe = {}
    for z in zs:  # quick probe: 16 MiB through each build on one thread
        r = harness.cpu_baseline(z.path, level, 1, 16, size, first_index, kind)
        probe[z.path] = r["pack_seconds"]
    zbest = min(zs, key=lambda z: probe[z.path])
    z15 = next((z for z in zs if z.version.startswith("1.5")), zbest)
    cores = usable_cores()
    one = harness.cpu_baseline(zbest.path, level, 1, sample, size, first_index, kind)
    # every thread gets 64 entries (at most 8192 in all: 8 GiB of 1 MiB entries + their frame buffers, allocated before the clock starts)
    many = harness.cpu_baseline(zbest.path, level, cores, max(min(64 * cores, 8192), cores), size, first_index, kind) if cores > 1 else None
    ref = harness.cpu_baseline(z15.path, level, cores, min(sample, 256), size, first_index, kind)  # ratio yardstick: the 1.5.x build
    one15 = harness.cpu_baseline(z15.path, level, 1, max(sample // 4, 16), size, first_index, kind) if z15.path != zbest.path else one
    out = {"value": one["bytes"] / one["pack_seconds"] / GIB, "unit": "GiB/s", "cores": 1, "kind": "port",
           "unpack_value": one["bytes"] / one["unpack_seconds"] / GIB,
           # the codec alone (the driver's BLAKE3 is the oracle's portable port; the reference's blake3 crate has SIMD code several times faster,
           # so the reference's true single-thread rate lies between `value` and this)
           "value_without_hash": one["bytes"] / max(one["pack_seconds"] - one["pack_hash_seconds"], 1e-9) / GIB,
           "unpack_value_without_hash": one["bytes"] / max(one["unpack_seconds"] - one["unpack_hash_seconds"], 1e-9) / GIB,
           # the same driver on the libzstd 1.5.x build (the ratio yardstick; the reference pins 1.5.5), one thread
           "libzstd_1_5": {"version": z15.version, "value": one15["bytes"] / one15["pack_seconds"] / GIB, "unpack_value": one15["bytes"] / one15["unpack_seconds"] / GIB,
                           "value_without_hash": one15["bytes"] / max(one15["pack_seconds"] - one15["pack_hash_seconds"], 1e-9) / GIB, "entries": one15["bytes"] // size},
           "sample": "%d x %d B corpus entries (kind %d) from index %d, level %d; C driver tests/support/cpu_baseline.c (one CCtx + session reset per "
                     "entry, decompressStream in 131075/131072-byte steps, oracle BLAKE3 port on both sides); %s; %d host cores usable by this process"
                     % (sample, size, kind, first_index, level, one["info"], cores),
           "ratio": one["bytes"] / one["compressed_bytes"]}
    if many:
        out["multithreaded"] = {"value": many["bytes"] / many["pack_seconds"] / GIB, "unpack_value": many["bytes"] / many["unpack_seconds"] / GIB,
                                "unit": "GiB/s", "cores": cores, "kind": "port",
                                "sample": "%d entries, %d threads, one context per thread" % (many["bytes"] // size, cores)}
    return out, (ref["compressed_bytes"], ref["bytes"] // size, "libzstd %s" % z15.version)


def lib_sha16(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


def src_sha16():
    """Hash of the kernel / engine sources the library is built from (zarc_amd/csrc + the ABI header): what ties a committed PMC profile to
    the code it was taken on -- unlike the binary's own hash it does not depend on where or when the library was compiled."""
    import glob
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "zarc_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "zarc_amd", "csrc", "*.h")) +
                   [os.path.join(ROOT, "zarc_amd", "csrc", "Makefile"), os.path.join(ROOT, "include", "zarc_gpu.h")])
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return h.hexdigest()[:16]


def workload(args, rank, world):
    """(sizes of this rank's entries, corpus index of each, kind, level, description) -- the global list dealt by the product's sharder."""
    from zarc_amd import shard
    import math
    import random
    if args.config == "c5":   # BASELINE configs[4] shape: 64 KiB .. 16 MiB log-uniform, kinds round-robin, level 3
        rnd = random.Random(5)
        target = int(args.gib * GIB) * world
        sizes, tot = [], 0
        while tot < target:
            s = int(65536 * 2 ** (rnd.random() * 8))
            sizes.append(s)
            tot += s
        level, kind = 3, -1
        desc = "zarc pack %d mixed entries 64 KiB..16 MiB (log-uniform, %.1f GiB), zstd level 3 (BASELINE configs[4] shape)" % (len(sizes), tot / GIB)
    elif args.config == "small":  # the reference's own published workload (README.md:288-329): 172 k files, median 822 B -- log-normal sizes
        rnd = random.Random(822)
        n = args.entries * world
        sizes = [max(1, min(16 << 20, int(math.exp(rnd.gauss(math.log(822.0), 1.819))))) for _ in range(n)]   # 95th percentile about 16 KiB
        level, kind = 3, -1
        desc = "zarc pack %d small entries per GPU (log-normal sizes, median 822 B, 95th percentile 16 KiB, %.2f GiB), zstd level 3 (shape of the reference's README benchmark tree)" % (
            args.entries, sum(sizes) / GIB / world)
    elif args.config == "c4":  # BASELINE configs[3] shape: one long lz stream cut into 4 MiB frames, level 9
        n = int(args.gib * GIB) // (4 << 20) * world
        sizes = [4 << 20] * n
        level, kind = 9, 2
        desc = "zarc pack %d x 4 MiB frames of the lz stream (%.1f GiB), zstd level 9 (BASELINE configs[3] shape)" % (n, n * 4 / 1024.0)
    else:
        n = args.entries * world
        sizes = [args.size] * n
        level, kind = 3, args.kind
        desc = "zarc pack %d x %d B synthetic entries per GPU, zstd level 3, checksum on (BASELINE configs[1]%s)" % (
            args.entries, args.size, "" if (args.entries, args.size) == (10000, 1 << 20) else ", scaled")
    if getattr(args, "level", None) is not None and args.level != level:
        desc = desc.replace("zstd level %d" % level, "zstd level %d (--level: not the configuration's %d)" % (args.level, level))
        level = args.level
    mine = shard.assign(sizes, world)[rank]          # positions in the global list
    if args.config == "c2":
        # Weak scaling keeps every rank's share the SAME mix of kinds: list position j holds corpus entry (j mod G) * entries + j // G, so
        # that index-mod-G dealing hands rank r the corpus range [r * entries, (r + 1) * entries) (kind = corpus index mod 4; dealing
        # corpus index j itself would give each of 4 or 8 ranks a single kind).
        cidx = [(j % world) * args.entries + j // world for j in mine]
    else:
        cidx = list(mine)
    return [sizes[i] for i in mine], cidx, kind, level, desc, len(sizes)


def host_path(eng, _lib, torch, d_src, off, lens, n_host, size):
    """PCIe-inclusive rates through zarc_gpu_pack_batch / zarc_gpu_unpack_batch (what a caller with ordinary buffers gets): the first
    n_host entries of the resident batch are copied to host memory first; only the ABI calls are timed."""
    out = {}
    nbytes = n_host * size
    cap = int(eng.bound(size)) * n_host
    for mode in ("pageable", "pinned"):
        pin = mode == "pinned"
        src = torch.empty(nbytes, dtype=torch.uint8, pin_memory=pin)
        dst = torch.empty(cap, dtype=torch.uint8, pin_memory=pin)
        back = torch.empty(nbytes, dtype=torch.uint8, pin_memory=pin)
        eng._check(eng.lib.zarc_gpu_memcpy_d2h(eng.h, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(d_src), nbytes))
        ptrs = (ctypes.c_void_p * n_host)(*[src.data_ptr() + i * size for i in range(n_host)])
        ln = (ctypes.c_size_t * n_host)(*[size] * n_host)
        doff, dlen = (ctypes.c_size_t * n_host)(), (ctypes.c_size_t * n_host)()
        dig, st = np.zeros((n_host, 32), dtype=np.uint8), (ctypes.c_int * n_host)()
        dig2 = np.zeros((n_host, 32), dtype=np.uint8)
        best = [1e30, 1e30]
        for _ in range(2):
            t0 = time.perf_counter()
            eng._check(eng.lib.zarc_gpu_pack_batch(eng.h, n_host, ptrs, ln, ctypes.c_void_p(dst.data_ptr()), cap, doff, dlen,
                                                   dig.ctypes.data_as(ctypes.c_void_p), st))
            t1 = time.perf_counter()
            fptrs = (ctypes.c_void_p * n_host)(*[dst.data_ptr() + doff[i] for i in range(n_host)])
            flens = (ctypes.c_size_t * n_host)(*[dlen[i] for i in range(n_host)])
            optrs = (ctypes.c_void_p * n_host)(*[back.data_ptr() + i * size for i in range(n_host)])
            t2 = time.perf_counter()
            eng._check(eng.lib.zarc_gpu_unpack_batch(eng.h, n_host, fptrs, flens, ln, optrs, dig.ctypes.data_as(ctypes.c_void_p),
                                                     dig2.ctypes.data_as(ctypes.c_void_p), st))
            t3 = time.perf_counter()
            best = [min(best[0], t1 - t0), min(best[1], t3 - t2)]
        assert all(s == 0 for s in st) and (dig2 == dig).all() and bool((back == src).all())
        out[mode] = {"pack_gibs": round(nbytes / best[0] / GIB, 2), "unpack_gibs": round(nbytes / best[1] / GIB, 2)}
        del src, dst, back
    out["bytes"] = nbytes
    out["note"] = "zarc_gpu_pack_batch / zarc_gpu_unpack_batch with host pointers, best of 2, H2D + kernels + D2H inside the timed call"
    return out


def launch_command(n, argv, port):
    """The command a plain `python bench.py --gpus N ...` starts as a child: one rank per GPU of this node over RCCL."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_ranks(n, argv):
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # this pool's driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(launch_command(n, argv, port), env=env)   # stdout / stderr are inherited: rank 0's JSON line goes straight out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=["c2", "c4", "c5", "small"], default="c2", help="c2: BASELINE configs[1]/[2] (the headline); c4 / c5: configs[3] / [4] shapes; small: a million entries of log-normal sizes, median 822 B")
    ap.add_argument("--gib", type=float, default=32.0, help="c4 / c5: uncompressed GiB per GPU")
    ap.add_argument("--entries", type=int, default=None, help="entries per GPU (default: BASELINE configs[1]'s 10000; --config small: 1000000)")
    ap.add_argument("--size", type=int, default=1 << 20, help="bytes per entry (BASELINE configs[1]: 1 MiB)")
    ap.add_argument("--cpu-sample", type=int, default=1024, help="entries timed on one host thread for cpu_baseline (rank 0, N=1)")
    ap.add_argument("--host-entries", type=int, default=8192, help="entries (of the resident batch) sent through the host-pointer entry points")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true")
    ap.add_argument("--dec-groups", type=int, default=0, help="diagnostics: ZARC_GPU_PX_DEC_GROUPS (0 = the engine decides by the batch's shape)")
    ap.add_argument("--level", type=int, default=None, help="compression level instead of the configuration's (c2 / c5 / small: 3, c4: 9); the metric then names it")
    ap.add_argument("--kind", type=int, default=-1, help="diagnostics: use one corpus kind for every entry (default: round-robin)")
    args = ap.parse_args()
    if args.entries is None:
        args.entries = 1000000 if args.config == "small" else 10000

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under a launcher: start the ranks ourselves, as a child (this process has not imported torch nor touched HIP)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.gpus != world:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE is %d (launcher and flag disagree)" % (args.gpus, world))
    import torch
    dist = None
    if world > 1 or "RANK" in os.environ:  # under torchrun always go through RCCL, even with one rank
        import torch.distributed as dist_mod
        torch.cuda.set_device(local_rank)
        dist_mod.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        dist = dist_mod
    device = torch.device("cuda", local_rank)

    from zarc_amd import Engine, _lib
    sizes, index, kind, level, desc, n_global = workload(args, rank, world)
    eng = Engine(local_rank)
    eng.set_parameter(_lib.P_COMPRESSION_LEVEL, level)
    eng.set_parameter(_lib.P_CHECKSUM_FLAG, 1)  # crates/zarc-cli/src/pack.rs:227
    if args.dec_groups:
        eng.set_parameter(_lib.PX_DEC_GROUPS, args.dec_groups)

    n = len(sizes)
    lens = np.array(sizes, dtype=np.uint64)
    strides = (lens + np.uint64(15)) // np.uint64(16) * np.uint64(16)
    off = np.concatenate([[0], np.cumsum(strides)[:-1]]).astype(np.uint64)
    total_in = int(strides.sum())
    bound_of = {s: int(eng.bound(int(s))) for s in set(sizes)}
    cap = sum(bound_of[s] for s in sizes)
    d_src = eng.malloc(total_in + _lib.PAD)
    d_dst = eng.malloc(cap + _lib.PAD)
    d_out = eng.malloc(total_in + _lib.PAD)
    # corpus entry of global index g is generated as corpus index g (any rank can generate any entry independently); c4 cuts ONE lz
    # stream per frame index the same way
    idx = np.array(index, dtype=np.int64)
    runs = np.split(np.arange(n), np.where(np.diff(idx) != 1)[0] + 1) if n else []
    for r in runs:  # contiguous index runs -> one fill call each (index mod G sharding gives runs of one; fill per entry then)
        eng.corpus_fill(d_src + int(off[r[0]]), off[r] - off[r[0]], lens[r], first_index=int(idx[r[0]]), kind=kind)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    # ---------------- pack ----------------
    for _ in range(args.warmup):
        doff, dlen, dig, st = eng.pack_device(d_src, off, lens, d_dst, cap)
    barrier()
    t0 = time.perf_counter()
    k_ms = np.zeros(_lib.T_DEC_FRAMES + 1)
    for _ in range(args.steps):
        doff, dlen, dig, st = eng.pack_device(d_src, off, lens, d_dst, cap)   # synchronous at the ABI
        k_ms += np.array([eng.kernel_ms(i) for i in range(_lib.T_DEC_FRAMES + 1)])
    barrier()
    t_pack = time.perf_counter() - t0
    assert (st == 0).all()
    k_ms /= max(args.steps, 1)
    raw_bytes = float(lens.sum())
    comp_bytes = float(dlen.sum())

    # ---------------- unpack ----------------
    for _ in range(args.warmup):
        dig2, st2 = eng.unpack_device(d_dst, doff, dlen, d_out, off, lens, expect=dig)
    barrier()
    t0 = time.perf_counter()
    u_ms = np.zeros(_lib.T_DEC_FRAMES + 1)
    for _ in range(args.steps):
        dig2, st2 = eng.unpack_device(d_dst, doff, dlen, d_out, off, lens, expect=dig)
        u_ms += np.array([eng.kernel_ms(i) for i in range(_lib.T_DEC_FRAMES + 1)])
    barrier()
    t_unpack = time.perf_counter() - t0
    u_ms /= max(args.steps, 1)
    ok = bool((st2 == 0).all() and (dig2 == dig).all())      # bit-exact round trip: same BLAKE3 on both sides

    tp, units = aggregate(t_pack, raw_bytes * args.steps, dist, device)
    tu, _ = aggregate(t_unpack, raw_bytes * args.steps, dist, device)
    pack_gibs = units / tp / GIB
    unpack_gibs = units / tu / GIB

    if rank == 0:
        # roofline of the dominant kernel of the headline (pack) path: the match finder.  Algorithmic bytes per
        # frame = N + C + 32; duration from HIP events on the engine's stream.
        names = {_lib.T_BLAKE3: "blake3", _lib.T_XXH64: "xxh64", _lib.T_MATCH: "zge_match", _lib.T_ENTROPY: "zge_entropy",
                 _lib.T_ASSEMBLE: "zge_assemble"}
        dom = max(names, key=lambda i: k_ms[i])
        alg_bytes = raw_bytes + comp_bytes + 32.0 * n
        achieved = alg_bytes / (k_ms[dom] * 1e-3) / 1e9 if k_ms[dom] > 0 else 0.0
        unames = {_lib.T_DEC_SEQS: "zdec_seqs", _lib.T_DEC_LITS: "zdec_literals", _lib.T_DEC_FRAMES: "zstd_frames", _lib.T_XXH64: "xxh64",
                  _lib.T_BLAKE3: "blake3"}
        udom = max(unames, key=lambda i: u_ms[i])
        u_achieved = alg_bytes / (u_ms[udom] * 1e-3) / 1e9 if u_ms[udom] > 0 else 0.0
        # HBM traffic of those kernels from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs;
        # FETCH_SIZE doubled per MI355X_MICROARCH.md -- calibrated on zarc_xxh64, which reads exactly N bytes and reports N/2).
        # Only quoted when the profile was taken on THESE kernel sources (hash recorded by tools/profile.sh) and this workload.
        traffic = u_traffic = None
        traffic_src = "none: no PMC profile of this library build under profiles/"
        pmc_file = os.path.join(ROOT, "profiles", "r04_bench_pmc_summary.json")
        if os.path.exists(pmc_file):
            pmc = json.load(open(pmc_file))
            same = pmc.get("src_sha16") == src_sha16() and pmc.get("entries") == n and pmc.get("entry_bytes") == args.size and args.config == "c2"
            if same:
                def tr(kname):
                    # (the sequence stage of unpack is two launches since round 4: the shared-table kernel and, for what it turns down, the old one)
                    knames = [k for k in pmc.get("fetch", {}) if k == kname or (kname == "zarc_zdec_seqs" and k.startswith("zarc_zdec_seqs"))]
                    if knames and all(k in pmc.get("write", {}) for k in knames):
                        return sum((2.0 * pmc["fetch"][k]["per_dispatch"] + pmc["write"][k]["per_dispatch"]) * 1024.0 for k in knames)
                    return None
                traffic, u_traffic = tr("zarc_" + names[dom]), tr("zarc_" + unames[udom])
                traffic_src = "profiles/r04_bench_pmc_summary.json (same kernel sources %s, same workload)" % pmc["src_sha16"]
            else:
                traffic_src = "none: profiles/r04_bench_pmc_summary.json is of another build or workload"
        line = {
            "metric": "uncompressed GiB/s (pack) at zstd -%d" % level, "value": round(pack_gibs, 3), "unit": "GiB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(tp / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": desc, "entries_per_gpu": n, "entries_total": n_global, "entry_bytes": args.size if args.config == "c2" else "mixed" if args.config in ("c5", "small") else 4 << 20,
                       "level": level, "kinds": "text/records/lz/random round-robin" if kind < 0 else "kind %d" % kind,
                       "parallelism": "frames dealt to %d GPU(s) by the product's sharder (index mod G / greedy by bytes), no collective" % world},
            "unpack_gibs": round(unpack_gibs, 3), "unpack_ms_per_step": round(tu / args.steps * 1e3, 3),
            "roundtrip_bit_exact": ok,
            "ratio": round(raw_bytes / comp_bytes, 4),
            "kernel_ms": {names[i]: round(float(k_ms[i]), 3) for i in names},
            "unpack_kernel_ms": {"zstd_decode": round(float(u_ms[_lib.T_DECODE]), 3), "zdec_seqs": round(float(u_ms[_lib.T_DEC_SEQS]), 3),
                                 "zdec_literals": round(float(u_ms[_lib.T_DEC_LITS]), 3), "zstd_frames": round(float(u_ms[_lib.T_DEC_FRAMES]), 3),
                                 "xxh64": round(float(u_ms[_lib.T_XXH64]), 3), "blake3": round(float(u_ms[_lib.T_BLAKE3]), 3)},
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(float(k_ms[dom]), 3)},
            "unpack_roofline": {"bound": "hbm", "kernel": unames[udom], "achieved": round(u_achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(u_achieved / HBM_PEAK_GBS, 5), "traffic": u_traffic,
                                "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(float(u_ms[udom]), 3)},
            "library_sha16": lib_sha16(eng.lib_path), "source_sha16": src_sha16(),
        }
        if world == 1 and args.config == "c2" and not args.no_host_path:
            nh = min(args.host_entries, n)
            line["host_path"] = host_path(eng, _lib, torch, d_src, off, lens, nh, args.size)
        if world == 1 and not args.no_cpu_baseline:
            size0 = int(lens[0])
            sample = min(args.cpu_sample, n) if args.config == "c2" else min(64, n)
            cb, ref = cpu_baseline(sample, {"c5": 1 << 20, "small": 4096}.get(args.config, size0), int(index[0]), level, kind)
            if cb:
                line["cpu_baseline"] = cb
                if args.config == "small":   # the ratio on the very entries of the batch: the first 20 000 through libzstd 1.5.x, entry by entry
                    sys.path.insert(0, os.path.join(ROOT, "tests", "support"))
                    import harness
                    z15 = next((z for z in harness.libzstds() if z.version.startswith("1.5")), None)
                    if z15:
                        m = min(20000, n)
                        corpus = harness.Corpus()
                        ref_comp = sum(len(z15.compress(corpus.entry(int(index[i]), int(lens[i]), kind), level, 1)) for i in range(m))
                        line["ratio_vs_reference"] = round(ref_comp / float(dlen[:m].sum()), 4)
                        line["ratio_reference"] = "libzstd %s -%d on the first %d entries, entry by entry" % (z15.version, level, m)
                if args.config in ("c2", "c4"):   # same entries on both sides (c5's sizes are mixed: its baseline sample is 1 MiB entries)
                    ref_comp, ref_n, ref_name = ref
                    ours = float(dlen[:ref_n].sum())
                    line["ratio_vs_reference"] = round(ref_comp / ours, 4)   # >= 0.95 required (within 5 % of libzstd -3)
                    line["ratio_reference"] = "%s -%d on the first %d entries" % (ref_name, level, ref_n)
        print(json.dumps(line))
    for p in (d_src, d_dst, d_out):
        eng.free(p)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()
    if not ok:
        sys.exit(1)


if __name__ == "__main__":
    main()
