"""Shared parity checks: the HIP path (through the C ABI) against the oracle.  Used twice: on the CPU build of
the same kernel sources under the HIP emulator (small inputs) and on a real MI355X (-m gpu)."""
import hashlib
import os

import numpy as np

import make_golden
from zarc_amd import _lib


def hash_cases(corpus, big):
    c = [b"", b"a", b"abc", bytes(i % 251 for i in range(1023)), bytes(i % 251 for i in range(1024)),
         bytes(i % 251 for i in range(1025)), bytes(i % 251 for i in range(2049)), bytes(i % 251 for i in range(31744)),
         corpus.entry(0, 65536, 0), corpus.entry(3, 70001, 3), corpus.entry(1, 131072 + 17, 1)]
    if big:
        c += [corpus.entry(2, 1 << 20, 2), corpus.entry(5, (1 << 20) + 1, 1), corpus.entry(7, 5 * (1 << 20) + 333, 3)]
    return c


def check_blake3(engine, oracle, corpus, big):
    ents = hash_cases(corpus, big)
    got = engine.blake3(ents)
    for e, d in zip(ents, got):
        assert d == oracle.blake3(e), len(e)


def check_xxh64_device(engine, oracle, corpus, big):
    ents = hash_cases(corpus, big)
    off, pos = [], 0
    for e in ents:
        off.append(pos)
        pos += (len(e) + 15) // 16 * 16
    d = engine.malloc(pos + _lib.PAD)
    try:
        for e, o in zip(ents, off):
            if e:
                engine.h2d(d + o, e)
        got = engine.xxh64_device(d, off, [len(e) for e in ents])
        for e, g in zip(ents, got):
            assert int(g) == oracle.xxh64(e), len(e)
    finally:
        engine.free(d)


def encode_cases(corpus, big):
    import random
    rnd = random.Random(11)
    c = {"empty": b"", "one": b"a", "abc": b"abc" * 9, "zeros": bytes(300000), "rand": corpus.entry(3, 70000, 3),
         "k0_1000": corpus.entry(0, 1000, 0), "k0_64k": corpus.entry(4, 65536, 0), "k1_64k": corpus.entry(5, 65536, 1),
         "k2_64k": corpus.entry(6, 65536, 2), "k0_200k": corpus.entry(8, 200000, 0), "k1_131073": corpus.entry(9, 131073, 1),
         # eighteen 64 KiB blocks = two table groups (sixteen blocks share their sequence tables: Repeat_Mode), RLE blocks and a stretch of
         # random bytes inside the first group (blocks that take no part / break the chain)
         "grp_10blk": corpus.entry(50, 3 * 131072, 2) + bytes(131072) + corpus.entry(51, 131072 + 70000, 3) + corpus.entry(52, 4 * 131072 - 70000 + 99, 0),
         "sparse": bytes(rnd.randrange(256) if i % 7 else 0 for i in range(100000)),
         "runs": make_golden.recipe_bytes({"kind": "runs", "n": 120000, "seed": 9}, corpus),
         "few": make_golden.recipe_bytes({"kind": "few", "n": 40000, "seed": 8}, corpus),
         # cold stretches of the match finder (unsearched tiles after tiles without a match) and the way back: random | text |
         # random | the same random again (a far repeat that starts inside an unsearched stretch) | records
         "cold_hot": corpus.entry(21, 40000, 3) + corpus.entry(22, 30000, 0) + corpus.entry(23, 50000, 3) + corpus.entry(23, 50000, 3)
                     + corpus.entry(24, 20000, 1),
         "cold_tail": corpus.entry(25, 9000, 0) + corpus.entry(26, 131072 + 5000, 3),   # stretch crosses a block boundary
         # one match per block once the 256-byte pieces are joined (runs of several hundred pieces: many rounds of the join pass)
         "periodic": bytes(range(200)) * 1500,
         # a repeat 120 KB back: long forgotten by the LDS tables, found through the far table in HBM; joined pieces across tiles
         "far_repeat": corpus.entry(27, 60000, 0) + corpus.entry(28, 60000, 1) + corpus.entry(27, 60000, 0) + corpus.entry(28, 30000, 1),
         "far_text": corpus.entry(29, 150000, 0) + corpus.entry(29, 150000, 0)[777:90000],
         # the 16-bit near table (round 3): several lanes of one 64-position group with the SAME bucket (short periods: the store is
         # contested and settled by the read-back), candidates exactly 65536 and 65537 bytes back (the 16-bit distance wraps), stale entries
         "per3": b"abc" * 4000, "per2": b"xy" * 3000 + b"q" + b"xy" * 3000,
         "per7n": bytes((i % 7) * 31 + (rnd.randrange(256) if rnd.randrange(40) == 0 else 0) & 255 for i in range(30000)),
         "wrap64k": corpus.entry(31, 100, 0) + corpus.entry(32, 1500, 1) + corpus.entry(33, 65536 - 1500, 0) + corpus.entry(32, 1500, 1)
                    + corpus.entry(34, 1, 0) + corpus.entry(33, 65536 - 1500, 0)[:30000] + corpus.entry(35, 65537 - 30000 - 1, 2) + corpus.entry(33, 65536 - 1500, 0)[:2000]}
    # found by tools/soak.py (round 3): a position whose near candidate is a hash collision at the very distance of its right neighbour's true
    # candidate -- the neighbour, a "follower", inherits length 0 and must still get the match through the far table's candidate
    c["follower_collision"] = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "enc_follower_collision.bin"), "rb").read()
    if big:
        for k in range(4):
            c["k%d_1m" % k] = corpus.entry(40 + k, 1 << 20, k)
        c["k1_2m5"] = corpus.entry(1, (5 << 20) // 2, 1)   # > 2^21: window-descriptor frame
        c["k0_4m"] = corpus.entry(12, 4 << 20, 0)
        c["periodic_4m"] = bytes(range(200)) * 20000   # crosses the 2 MiB table segment: the recent-offset guess carries the match over
        c["far_2m3"] = corpus.entry(13, 1 << 20, 0) + corpus.entry(14, 300000, 2) + corpus.entry(13, 1 << 20, 0)
    return c


def check_pack(engine, oracle, corpus, libzstds, big):
    """Frames are bit-identical to the encoder model, valid for the oracle decoder and for real libzstd."""
    assert libzstds, "no libzstd on this box: the cross-decoding half of this check would be vacuous"
    cases = encode_cases(corpus, big)
    names = list(cases)
    res = engine.pack([cases[k] for k in names])
    for k, (frame, dig) in zip(names, res):
        raw = cases[k]
        assert dig == oracle.blake3(raw), k
        assert frame == oracle.zge_encode(raw), k                      # bit-exact vs the CPU model
        rc, out, used = oracle.zstd_decode(frame, len(raw))
        assert rc == 0 and used == len(frame) and out == raw, k       # valid Zstandard
        for z in libzstds:
            got, err = z.decompress(frame, len(raw))
            assert got == raw, (k, z.version, err)
        assert len(frame) <= engine.bound(len(raw))


def check_levels(engine, oracle, corpus, libzstds, big):
    """The level tiers (zarc-cli/src/pack.rs:24-33 forwards --level -131072..22): level 1 and the negative levels run the fast finder (near
    table only), 2..8 the level-3 finder, 9..14 the deep finder, 15..22 the deep finder with two more live rounds.  Every tier bit-exact
    against the model at the same level, valid for the oracle decoder and libzstd; levels of one tier give the same bytes."""
    cases = encode_cases(corpus, big)
    names = [k for k in ("k0_64k", "k1_64k", "k0_200k", "far_repeat", "per3", "abc", "empty") if k in cases] + (["cold_hot", "k0_4m", "far_2m3"] if big else [])
    frames = {}
    try:
        for level in ((1, -1, -7, 2, 9, 15, 22) if big else (1, -7, 2, 9, 15)):
            engine.set_parameter(_lib.P_COMPRESSION_LEVEL, level)
            res = engine.pack([cases[k] for k in names])
            frames[level] = [f for f, _ in res]
            for k, (frame, dig) in zip(names, res):
                raw = cases[k]
                assert frame == oracle.zge_encode(raw, oracle.params(level=level)), (k, level)
                rc, out, used = oracle.zstd_decode(frame, len(raw))
                assert rc == 0 and used == len(frame) and out == raw, (k, level)
                for z in libzstds:
                    got, err = z.decompress(frame, len(raw))
                    assert got == raw, (k, level, z.version, err)
        assert frames[1] == frames[-7] and frames[1] != frames[2] and frames[2] != frames[9]
        if big:
            assert frames[1] == frames[-1] and frames[15] == frames[22]
    finally:
        engine.set_parameter(_lib.P_COMPRESSION_LEVEL, 3)


def check_unpack_golden(engine, oracle, corpus, golden_frames, limit=None):
    """Frames made by real libzstd builds decode bit-exactly; digest + checksum verified."""
    d, m = golden_frames
    frames, raws, cache = [], [], {}
    sel = m["frames"] if limit is None else [f for f in m["frames"] if f["raw_len"] <= limit]
    for fr in sel:
        name = fr["recipe"]
        if name not in cache:
            cache[name] = make_golden.recipe_bytes(m["recipes"][name], corpus)
        frames.append(open(os.path.join(d, fr["file"]), "rb").read())
        raws.append(cache[name])
    res = engine.unpack(frames, [len(r) for r in raws], [oracle.blake3(r) for r in raws])
    for fr, raw, (out, dig, st) in zip(sel, raws, res):
        assert st == _lib.FRAME_OK, (fr["file"], st)
        assert out == raw and dig == oracle.blake3(raw), fr["file"]
        assert hashlib.sha256(out).hexdigest() == fr["raw_sha256"]


def check_roundtrip(engine, oracle, corpus, big):
    cases = encode_cases(corpus, big)
    names = list(cases)
    packed = engine.pack([cases[k] for k in names])
    res = engine.unpack([p[0] for p in packed], [len(cases[k]) for k in names], [p[1] for p in packed])
    for k, (out, dig, st) in zip(names, res):
        assert st == _lib.FRAME_OK and out == cases[k], k


def check_unpack_errors(engine, oracle, corpus, golden_frames):
    """Per-frame status without aborting the batch (ok / checksum / digest / corrupt)."""
    d, m = golden_frames
    fr = next(f for f in m["frames"] if f["recipe"] == "text300" and f["level"] == 3 and f["checksum"] == 1 and f["libzstd"].startswith("1.5"))
    good = open(os.path.join(d, fr["file"]), "rb").read()
    raw = make_golden.recipe_bytes(m["recipes"]["text300"], corpus)
    bad_ck = bytearray(good); bad_ck[-1] ^= 0x40
    bad_magic = bytearray(good); bad_magic[0] ^= 1
    trunc = good[:-7]
    corrupt = bytearray(good); corrupt[12] ^= 0xFF; corrupt[13] ^= 0xFF
    wrong_digest = bytes(32)
    frames = [good, bytes(bad_ck), bytes(bad_magic), trunc, bytes(corrupt), good, good]
    expect = [oracle.blake3(raw)] * 5 + [wrong_digest, oracle.blake3(raw)]
    raw_lens = [len(raw)] * 6 + [len(raw) + 1]
    res = engine.unpack(frames, raw_lens, expect)
    st = [r[2] for r in res]
    assert st[0] == _lib.FRAME_OK and res[0][0] == raw
    assert st[1] == _lib.FRAME_CHECKSUM
    assert st[2] == _lib.FRAME_BAD_MAGIC
    assert st[3] != _lib.FRAME_OK
    assert st[4] != _lib.FRAME_OK
    assert st[5] == _lib.FRAME_DIGEST and res[5][0] == raw       # reported, bytes still delivered (unpack.rs:118-120)
    assert st[6] == _lib.FRAME_SRCSIZE                            # Frame.uncompressed disagrees with the frame
    # the oracle agrees on which frames are undecodable
    for i in (2, 3, 4):
        assert oracle.zstd_decode(frames[i], len(raw))[0] != 0


def check_many_frames_with_turned_down_ones(engine, oracle, corpus, n):
    """A batch of many small frames in which some are not the fast path's: the general decoder (zarc_zstd_decode) then takes 64 queue slots
    per trip and decodes exactly the frames whose flag says so (zstd_decode.hip).  Every status and every byte as for the frame alone."""
    import random
    rnd = random.Random(11)
    ents = [corpus.entry(5000 + i, rnd.choice((0, 1, 9, 40, 130, 700, 2500)), i % 4) for i in range(n)]
    packed = engine.pack(ents)
    frames = [f for f, _ in packed]
    digests = [d for _, d in packed]
    raw_lens = [len(e) for e in ents]
    bad = {}
    for i in range(3, n, 37):            # scattered: several per 64-slot trip, and trips without any
        kind = (i // 37) % 4
        f = bytearray(frames[i])
        if kind == 0: f[0] ^= 1; bad[i] = "magic"
        elif kind == 1 and len(f) > 9: f = f[:-3]; bad[i] = "trunc"
        elif kind == 2: raw_lens[i] += 1; bad[i] = "size"
        elif kind == 3: digests[i] = bytes(32); bad[i] = "digest"
        frames[i] = bytes(f)
    res = engine.unpack(frames, raw_lens, digests)
    for i, (out, dig, st) in enumerate(res):
        why = bad.get(i)
        if why is None: assert st == _lib.FRAME_OK and out == ents[i], i
        elif why == "magic": assert st == _lib.FRAME_BAD_MAGIC, i
        elif why == "trunc": assert st != _lib.FRAME_OK, i
        elif why == "size": assert st == _lib.FRAME_SRCSIZE, i
        elif why == "digest": assert (st == _lib.FRAME_DIGEST and out == ents[i]) or len(ents[i]) == 0 and st in (_lib.FRAME_DIGEST, _lib.FRAME_OK), i
    # the same frames one call each give the same statuses (a sample)
    for i in sorted(bad)[:12]:
        one = engine.unpack([frames[i]], [raw_lens[i]], [digests[i]])[0]
        assert one[2] == res[i][2], (i, bad[i])


def check_store(engine, oracle, corpus, libzstds):
    """Encoder::enable_compression(false) (encode.rs:95-97, lowlevel_frames.rs:47-84): raw-block frames.  The layout is pinned
    byte for byte: the reference's descriptor (8-byte content size, no single segment, no checksum) plus the window byte
    the reference forgets, 128 KiB raw blocks.  Every real libzstd must decode them."""
    sizes = [0, 1, 300, 65535, 65536, 131072, 131073, 400000]
    ents = [corpus.entry(700 + i, n, i & 3) for i, n in enumerate(sizes)]
    engine.enable_compression(False)
    try:
        res = engine.pack(ents)
    finally:
        engine.enable_compression(True)
    for (frame, dig), raw in zip(res, ents):
        want = bytes.fromhex("28B52FFD" "C0" "38") + len(raw).to_bytes(8, "little")
        nb = max(1, (len(raw) + 131071) // 131072)
        for b in range(nb):
            chunk = raw[b * 131072:(b + 1) * 131072]
            want += ((1 if b + 1 == nb else 0) | (len(chunk) << 3)).to_bytes(3, "little") + chunk
        assert frame == want
        assert dig == oracle.blake3(raw)
        st, out, _ = oracle.zstd_decode(frame, len(raw))
        assert st == 0 and out == raw
        for z in libzstds:
            out, err = z.decompress(frame, len(raw))
            assert err is None and out == raw, (z.version, err)
    outs = engine.unpack([f for f, _ in res], [len(e) for e in ents], expect=[d for _, d in res])
    for (out, dig, st), raw in zip(outs, ents):
        assert st == 0 and out == raw


def check_params(engine):
    import pytest
    from zarc_amd import ZarcGpuError
    engine.set_parameter(_lib.P_COMPRESSION_LEVEL, 3)
    engine.set_parameter(_lib.P_WINDOW_LOG, 0)
    with pytest.raises(ZarcGpuError):
        engine.set_parameter(_lib.P_COMPRESSION_LEVEL, 23)
    with pytest.raises(ZarcGpuError):
        engine.set_parameter(400, 4)      # NbWorkers: known to libzstd, unsupported here
    # search-effort hints: accepted inside libzstd's bounds and remembered (the reference forwards them all, pack.rs:86-217), advisory
    for pid, good, bad, field in ((_lib.P_HASH_LOG, 20, 31, "hash_log"), (_lib.P_CHAIN_LOG, 16, 5, "chain_log"), (_lib.P_SEARCH_LOG, 4, 31, "search_log"),
                                  (_lib.P_TARGET_LENGTH, 64, 131073, "target_length"), (_lib.P_STRATEGY, 7, 10, "strategy")):
        engine.set_parameter(pid, good)
        assert getattr(engine.params(), field) == good
        with pytest.raises(ZarcGpuError) as ei:
            engine.set_parameter(pid, bad)
        assert ei.value.code == _lib.E_PARAM
        engine.set_parameter(pid, 0)      # back to "use the default"
        assert getattr(engine.params(), field) == 0
    # long-distance matching: accepted inside libzstd's bounds, advisory (the far tables are the engine's long-distance matcher)
    for pid, good, bad in ((_lib.P_ENABLE_LDM, 1, 3), (_lib.P_LDM_HASH_LOG, 20, 31), (_lib.P_LDM_MIN_MATCH, 64, 3), (_lib.P_LDM_BUCKET_SIZE_LOG, 3, 9),
                           (_lib.P_LDM_HASH_RATE_LOG, 7, 26)):
        engine.set_parameter(pid, good)
        assert engine.lib.zarc_gpu_parameter_advisory(pid) == 1
        with pytest.raises(ZarcGpuError) as ei:
            engine.set_parameter(pid, bad)
        assert ei.value.code == _lib.E_PARAM
        engine.set_parameter(pid, 0)
    assert engine.lib.zarc_gpu_parameter_advisory(_lib.P_STRATEGY) == 1 and engine.lib.zarc_gpu_parameter_advisory(_lib.P_WINDOW_LOG) == 0
    assert [engine.lib.zarc_gpu_level_finder(l) for l in (-131072, -1, 0, 1, 2, 3, 8, 9, 14, 15, 22)] == [1, 1, 3, 1, 3, 3, 3, 9, 9, 15, 15]
    with pytest.raises(ZarcGpuError):
        engine.set_parameter(31337, 1)    # unknown id
    assert engine.params().checksum_flag == 1


def check_unpack_fuzz(engine, oracle, corpus, golden_frames, n_mut, seed, max_raw=140000):
    """Mutated frames (bit flips, byte smashes, truncations, header edits): the HIP decoder and the oracle decoder
    must agree on whether the frame is decodable, and on every output byte when it is.  Nothing may hang or fault."""
    import random
    d, m = golden_frames
    rnd = random.Random(seed)
    pool = [f for f in m["frames"] if 0 < f["raw_len"] <= max_raw]
    frames, raw_lens, want = [], [], []
    while len(frames) < n_mut:
        fr = rnd.choice(pool)
        data = bytearray(open(os.path.join(d, fr["file"]), "rb").read())
        if fr["checksum"] and rnd.randrange(2):     # half of the time without the XXH64 trailer, so that damage is not
            data[4] &= ~0x04                        # simply caught by the checksum and the decoders must agree on
            del data[-4:]                           # structure (or on the garbage they both produce)
        kind = rnd.randrange(6)
        if kind == 0 and len(data) > 12:            # flip one bit somewhere after the magic
            i = rnd.randrange(4, len(data)); data[i] ^= 1 << rnd.randrange(8)
        elif kind == 1 and len(data) > 12:          # smash a byte
            data[rnd.randrange(4, len(data))] = rnd.randrange(256)
        elif kind == 2 and len(data) > 12:          # truncate
            data = data[:rnd.randrange(5, len(data))]
        elif kind == 3:                             # extend with garbage
            data += bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 9)))
        elif kind == 4 and len(data) > 12:          # damage a block header / section header region
            i = rnd.randrange(5, min(len(data), 40)); data[i] ^= rnd.randrange(1, 256)
        else:                                       # several random flips
            for _ in range(rnd.randrange(2, 6)):
                i = rnd.randrange(4, len(data)); data[i] ^= 1 << rnd.randrange(8)
        raw_len = fr["raw_len"] if rnd.randrange(8) else max(0, fr["raw_len"] + rnd.choice((-1, 1, 100)))
        frames.append(bytes(data)); raw_lens.append(raw_len)
        rc, out, used = oracle.zstd_decode(bytes(data), raw_len)
        want.append(out if (rc == 0 and used == len(data) and len(out) == raw_len) else None)
    res = engine.unpack(frames, raw_lens)
    agree_ok = agree_bad = 0
    for i, ((out, dig, st), w) in enumerate(zip(res, want)):
        if w is None:
            assert st != _lib.FRAME_OK, ("engine accepted a frame the oracle rejects", i, len(frames[i]), raw_lens[i])
            agree_bad += 1
        else:
            assert st == _lib.FRAME_OK and out == w, ("engine rejected/garbled a frame the oracle decodes", i, st)
            agree_ok += 1
    return agree_ok, agree_bad
