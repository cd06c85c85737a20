#!/usr/bin/env python3
"""Unpack frames made by libzstd from a cache file (so that the run itself never calls libzstd: rocprofv3's preloaded tool brings its own
ZSTD_* symbols).  usage: libzstd_frames_cache.py make <file> [entries] [level]   |   libzstd_frames_cache.py run <file> [repeats]"""
import os, sys, time, pickle
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "support"))
size = 1 << 20
if sys.argv[1] == "make":
    from concurrent.futures import ThreadPoolExecutor
    import harness
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
    lv = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    corpus = harness.Corpus()
    z = next(z for z in harness.libzstds() if z.version.startswith("1.5"))
    ents = [corpus.entry(i, size, -1) for i in range(n)]
    with ThreadPoolExecutor(16) as ex:
        frames = list(ex.map(lambda e: z.compress(e, lv, 1), ents))
    pickle.dump({"frames": frames, "check": ents[5]}, open(sys.argv[2], "wb"))
    print("wrote", sys.argv[2], len(frames), "frames")
else:
    from zarc_amd import Engine, _lib
    d = pickle.load(open(sys.argv[2], "rb"))
    frames, n = d["frames"], len(d["frames"])
    eng = Engine(0); eng.set_parameter(_lib.P_CHECKSUM_FLAG, 1)
    off, pos = [], 0
    for f in frames:
        off.append(pos); pos += (len(f) + 15) // 16 * 16
    blob = np.zeros(pos + _lib.PAD, dtype=np.uint8)
    for f, o in zip(frames, off): blob[o:o + len(f)] = np.frombuffer(f, dtype=np.uint8)
    d_fr, d_out = eng.malloc(len(blob)), eng.malloc(n * size + _lib.PAD)
    eng.h2d(d_fr, blob)
    doff = [i * size for i in range(n)]
    best = 1e9
    for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 3):
        t0 = time.perf_counter()
        dig, st = eng.unpack_device(d_fr, off, [len(f) for f in frames], d_out, doff, [size] * n)
        best = min(best, time.perf_counter() - t0)
    assert (st == 0).all()
    assert bytes(eng.d2h(d_out + 5 * size, size)) == d["check"]
    print("%d libzstd frames: %.1f GiB/s  decode %.1f ms (seqs %.1f, literals %.1f, frame pass %.1f)" % (
        n, n * size / best / 2**30, eng.kernel_ms(_lib.T_DECODE), eng.kernel_ms(_lib.T_DEC_SEQS), eng.kernel_ms(_lib.T_DEC_LITS), eng.kernel_ms(_lib.T_DEC_FRAMES)))
