#!/usr/bin/env python3
"""How libzstd's OWN ratio moves when its tables shrink to the sizes that fit LDS, when its chain search gets shallower, and when
its strategy drops to greedy / lazy -- on the real-data items of tests/support/realdata.py.  Evidence for DESIGN.md section 4.1: on
machine code the level-9 advantage is the parse policy (lazy2 + live repeat offsets), not table size."""
import sys, ctypes
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "support"))
import harness, realdata, glob
z = next(z for z in harness.libzstds() if z.version.startswith("1.5"))
items = {k: v for k, v in realdata.items().items() if v}
def comp(d, level, **p):
    Z = z.z
    c = Z.ZSTD_createCCtx()
    Z.ZSTD_CCtx_setParameter(c, 100, level); Z.ZSTD_CCtx_setParameter(c, 201, 1)
    ids = dict(windowLog=101, hashLog=102, chainLog=103, searchLog=104, minMatch=105, targetLength=106, strategy=107)
    for k, v in p.items():
        r = Z.ZSTD_CCtx_setParameter(c, ids[k], v); assert not Z.ZSTD_isError(r), k
    cap = len(d) + len(d) // 8 + 1024
    dst = ctypes.create_string_buffer(cap)
    n = Z.ZSTD_compress2(c, dst, cap, d, len(d)); Z.ZSTD_freeCCtx(c)
    assert not Z.ZSTD_isError(n)
    return n
for n, d in items.items():
    base3 = comp(d, 3); base9 = comp(d, 9)
    print("%-16s L3 %8d | h13c13 %.3f | h15c15 %.3f | L9 %8d | L9 h14c14 %.3f | L9 h16c16 %.3f | L9 s1 %.3f | L9 s2 %.3f| greedy(3) %.3f lazy(4) %.3f" % (n, base3, comp(d, 3, hashLog=13, chainLog=13) / base3, comp(d, 3, hashLog=15, chainLog=15) / base3,
          base9, comp(d, 9, hashLog=14, chainLog=14) / base9, comp(d, 9, hashLog=16, chainLog=16) / base9, comp(d, 9, searchLog=1) / base9, comp(d, 9, searchLog=2) / base9, comp(d, 9, strategy=3)/base9, comp(d, 9, strategy=4)/base9))
