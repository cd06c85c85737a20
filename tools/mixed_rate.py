#!/usr/bin/env python3
"""BASELINE configs[4] shape on one GPU: entries of 64 KiB .. 16 MiB (log-uniform), kinds round-robin, level 3, resident in HBM.
usage: mixed_rate.py [GiB of content, default 24] [seed]"""
import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zarc_amd import Engine, _lib

target = float(sys.argv[1]) if len(sys.argv) > 1 else 24.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
sizes = []
while sum(sizes) < target * 2**30:
    sizes.append(int(65536 * 2.0 ** (rng.random() * 8)))      # 65 536 * 2^(u*8), u ~ U[0,1): SURVEY.md section 8(d), C5
lens = np.array(sizes, dtype=np.uint64)
stride = (lens + np.uint64(15)) // np.uint64(16) * np.uint64(16)
off = np.concatenate([[0], np.cumsum(stride)[:-1]]).astype(np.uint64)
total = int(stride.sum())
eng = Engine(0); eng.set_parameter(_lib.P_COMPRESSION_LEVEL, 3); eng.set_parameter(_lib.P_CHECKSUM_FLAG, 1)
cap = int(sum(int(eng.bound(int(x))) for x in sizes))
d_src = eng.malloc(total + _lib.PAD); d_dst = eng.malloc(cap + _lib.PAD); d_out = eng.malloc(total + _lib.PAD)
eng.corpus_fill(d_src, off, lens, first_index=0, kind=-1)
best = [1e9, 1e9]
for rep in range(3):
    t0 = time.perf_counter(); doff, dlen, dig, st = eng.pack_device(d_src, off, lens, d_dst, cap); t1 = time.perf_counter()
    dig2, st2 = eng.unpack_device(d_dst, doff, dlen, d_out, off, lens, expect=dig); t2 = time.perf_counter()
    best = [min(best[0], t1 - t0), min(best[1], t2 - t1)]
ok = bool((st == 0).all() and (st2 == 0).all() and (dig2 == dig).all())
raw = float(lens.sum())
print("mixed sizes 64 KiB..16 MiB: %d entries, %.1f GiB: pack %.2f GiB/s, unpack %.2f GiB/s, ratio %.4f, round trip %s, match %.1f ms entropy %.1f ms"
      % (len(sizes), raw / 2**30, raw / best[0] / 2**30, raw / best[1] / 2**30, raw / float(dlen.sum()), ok,
         eng.kernel_ms(_lib.T_MATCH), eng.kernel_ms(_lib.T_ENTROPY)))
