#!/usr/bin/env python3
"""Unpack rate on frames made by libzstd (what an archive written by the reference holds) next to the engine's own frames of the same
entries, both resident in HBM.  usage: libzstd_frames_rate.py [entries] [levels...]"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "support"))
from zarc_amd import Engine, _lib
import harness

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
levels = [int(x) for x in sys.argv[2:]] or [3, 9, 19]
size = 1 << 20
corpus, oracle = harness.Corpus(), harness.Oracle()
z = next(z for z in harness.libzstds() if z.version.startswith("1.5"))
ents = [corpus.entry(i, size, -1) for i in range(n)]
eng = Engine(0); eng.set_parameter(_lib.P_CHECKSUM_FLAG, 1)

def run(frames, label):
    off, pos = [], 0
    for f in frames:
        off.append(pos); pos += (len(f) + 15) // 16 * 16
    blob = np.zeros(pos + _lib.PAD, dtype=np.uint8)
    for f, o in zip(frames, off): blob[o:o + len(f)] = np.frombuffer(f, dtype=np.uint8)
    d_fr, d_out = eng.malloc(len(blob)), eng.malloc(n * size + _lib.PAD)
    eng.h2d(d_fr, blob)
    doff = [i * size for i in range(n)]
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        dig, st = eng.unpack_device(d_fr, off, [len(f) for f in frames], d_out, doff, [size] * n)
        best = min(best, time.perf_counter() - t0)
    assert (st == 0).all(), label
    out = eng.d2h(d_out + 5 * size, size)
    assert bytes(out) == ents[5]
    print("%-22s %6.1f GiB/s  decode %.1f ms (seqs %.1f, literals %.1f, frame pass %.1f)  compressed %.1f MiB" % (
        label, n * size / best / 2**30, eng.kernel_ms(_lib.T_DECODE), eng.kernel_ms(_lib.T_DEC_SEQS), eng.kernel_ms(_lib.T_DEC_LITS),
        eng.kernel_ms(_lib.T_DEC_FRAMES), sum(len(f) for f in frames) / 2**20), flush=True)
    eng.free(d_fr); eng.free(d_out)

run([f for f, _ in eng.pack(ents)], "engine frames (L3)")
for lv in levels:
    if os.environ.get("ZARC_TOOL_THREADS") == "1": frames = [z.compress(e, lv, 1) for e in ents]   # (under rocprofv3: its preloaded tool does not survive the pool)
    else:
        with ThreadPoolExecutor(16) as ex:
            frames = list(ex.map(lambda e: z.compress(e, lv, 1), ents))
    run(frames, "libzstd %s -%d" % (z.version, lv))
