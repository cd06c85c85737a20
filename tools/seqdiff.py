#!/usr/bin/env python3
"""Which sequence do two Zstandard frames of the same data first disagree on?  (debugging aid: GPU / emulator frame against the model's)
usage: seqdiff.py input.bin level [emu]   -- packs input with the emulator build (or the GPU library) and with the model, lists the first differences"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "support"))
import harness
from zarc_amd import Engine, _lib
raw = open(sys.argv[1], "rb").read()
level = int(sys.argv[2])
o = harness.Oracle()
lib = os.path.join(ROOT, "tests", "emu", "_build", "libzarc_gpu_emu.so") if len(sys.argv) > 3 else None
e = Engine(0, lib)
e.set_parameter(_lib.P_COMPRESSION_LEVEL, level); e.set_parameter(_lib.P_CHECKSUM_FLAG, 0)
(frame, _), = e.pack([raw])
model = o.zge_encode(raw, o.params(level=level, checksum=0))
TR = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32)
def trace(f):
    seqs = []
    cb = TR(lambda ctx, pos, ll, ml, off: seqs.append((pos, ll, ml, off)))
    ctypes.c_void_p.in_dll(o.lib, "oracle_zstd_trace").value = ctypes.cast(cb, ctypes.c_void_p).value
    rc, out, used = o.zstd_decode(f, len(raw))
    ctypes.c_void_p.in_dll(o.lib, "oracle_zstd_trace").value = None
    assert rc == 0 and out == raw
    return seqs
a, b = trace(frame), trace(model)
print("engine: %d bytes, %d sequences; model: %d bytes, %d sequences" % (len(frame), len(a), len(model), len(b)))
for i, (x, y) in enumerate(zip(a, b)):
    if x != y:
        print("first difference at sequence %d: engine (pos, ll, ml, off) = %s, model = %s" % (i, x, y))
        for j in range(max(0, i - 3), min(len(a), len(b), i + 4)): print("   ", j, a[j], b[j])
        break
if len(a) != len(b) and a[:min(len(a), len(b))] == b[:min(len(a), len(b))]:
    print("one list is a prefix of the other; tails:"); print("  engine", a[-3:]); print("  model ", b[-3:])
