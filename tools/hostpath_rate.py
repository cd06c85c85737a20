#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry points (zarc_gpu_pack_batch / zarc_gpu_unpack_batch): only the ABI calls are
timed; buffers are ordinary pageable host memory, as a caller of the reference library would hand over."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "support"))
from zarc_amd import Engine, _lib
import harness

n, size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 1 << 20
corpus = harness.Corpus()
eng = Engine(0); eng.set_parameter(_lib.P_COMPRESSION_LEVEL, 3); eng.set_parameter(_lib.P_CHECKSUM_FLAG, 1)
PINNED = "pinned" in sys.argv[2:]
def buf(nbytes):
    """numpy view of nbytes of host memory: ordinary, or page-locked by the HIP runtime the engine itself uses (hipHostMalloc)"""
    if not PINNED:
        return np.zeros(nbytes, dtype=np.uint8)
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
    p = ctypes.c_void_p()
    assert hip.hipHostMalloc(ctypes.byref(p), nbytes, 0) == 0
    return np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(p.value))
src = buf(n * size)
for i in range(n):
    src[i * size:(i + 1) * size] = np.frombuffer(corpus.entry(i % 512, size, -1), dtype=np.uint8)
ptrs = (ctypes.c_void_p * n)(*[src.ctypes.data + i * size for i in range(n)])
lens = (ctypes.c_size_t * n)(*[size] * n)
cap = eng.bound(size) * n
dst = buf(cap)
doff, dlen = (ctypes.c_size_t * n)(), (ctypes.c_size_t * n)()
dig, st = np.zeros((n, 32), dtype=np.uint8), (ctypes.c_int * n)()
out = buf(n * size)
optrs = (ctypes.c_void_p * n)(*[out.ctypes.data + i * size for i in range(n)])
dig2 = np.zeros((n, 32), dtype=np.uint8)
# optional: PX id=value pairs after the entry count, e.g. hostpath_rate.py 8192 9004=16 9002=1073741824
for kv in sys.argv[2:]:
    if kv == "pinned":
        continue
    if kv == "store":   # --store frames: the kernels do next to nothing, the rate is the staging's own (N bytes in, N bytes out)
        eng.lib.zarc_gpu_enable_compression(eng.h, 0)
        continue
    k, v = kv.split("=")
    eng.set_parameter(int(k), int(v))
best = [1e9, 1e9]
for rep in range(3):
    t0 = time.perf_counter()
    eng._check(eng.lib.zarc_gpu_pack_batch(eng.h, n, ptrs, lens, dst.ctypes.data_as(ctypes.c_void_p), cap, doff, dlen, dig.ctypes.data_as(ctypes.c_void_p), st))
    t1 = time.perf_counter()
    fptrs = (ctypes.c_void_p * n)(*[dst.ctypes.data + doff[i] for i in range(n)])
    flens = (ctypes.c_size_t * n)(*[dlen[i] for i in range(n)])
    t2 = time.perf_counter()
    eng._check(eng.lib.zarc_gpu_unpack_batch(eng.h, n, fptrs, flens, lens, optrs, dig.ctypes.data_as(ctypes.c_void_p), dig2.ctypes.data_as(ctypes.c_void_p), st))
    t3 = time.perf_counter()
    best = [min(best[0], t1 - t0), min(best[1], t3 - t2)]
assert (out == src).all() and all(s == 0 for s in st)
print("host-pointer entry points, %d x %d B, host buffers (pageable unless 'pinned' follows) %s: pack %.2f GiB/s, unpack %.2f GiB/s" % (n, size, " ".join(sys.argv[2:]), n * size / best[0] / 2**30, n * size / best[1] / 2**30))
