"""Thin Python view of the engine handle (include/zarc_gpu.h).  Plumbing only: every byte of the data path
is processed by the HIP kernels inside libzarc_gpu.so."""
import ctypes

import numpy as np

from . import _lib
from ._lib import ZarcGpuError


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


class Engine:
    """One engine handle = one HIP stream on one device (CCtx/DCtx analogue, crates/zarc/src/encode.rs:58-78)."""

    def __init__(self, device=0, lib_path=None):
        import os
        self.lib_path = lib_path or os.environ.get("ZARC_GPU_LIB") or _lib.DEFAULT_LIB
        self.lib = _lib.load(self.lib_path)
        h = ctypes.c_void_p()
        rc = self.lib.zarc_gpu_create(ctypes.byref(h), device)
        if rc != 0:
            raise ZarcGpuError(rc, self.lib.zarc_gpu_error_name(rc).decode(),
                               "zarc_gpu_create failed: no usable HIP device (there is no CPU fallback)")
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.zarc_gpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise ZarcGpuError(rc, self.lib.zarc_gpu_error_name(rc).decode(), self.lib.zarc_gpu_last_error(self.h).decode())

    # ---- parameters (Encoder::set_zstd_parameter / enable_compression) ----
    def set_parameter(self, param_id, value):
        self._check(self.lib.zarc_gpu_set_parameter(self.h, param_id, value))

    def enable_compression(self, compress):
        """Encoder::enable_compression (crates/zarc/src/encode.rs:95-97): False -> raw-block ("stored") frames."""
        self.lib.zarc_gpu_enable_compression(self.h, 1 if compress else 0)

    def params(self):
        p = _lib.Params()
        self.lib.zarc_gpu_get_params(self.h, ctypes.byref(p))
        return p

    def bound(self, n):
        return self.lib.zarc_gpu_bound(n)

    def kernel_ms(self, which):
        return float(self.lib.zarc_gpu_last_kernel_ms(self.h, which))

    # ---- device memory helpers ----
    def malloc(self, nbytes):
        p = ctypes.c_void_p()
        self._check(self.lib.zarc_gpu_device_malloc(self.h, ctypes.byref(p), nbytes))
        return p.value

    def free(self, dptr):
        self._check(self.lib.zarc_gpu_device_free(self.h, ctypes.c_void_p(dptr)))

    def h2d(self, dptr, data):
        buf = (ctypes.c_char * len(data)).from_buffer_copy(data) if not isinstance(data, np.ndarray) else None
        src = buf if buf is not None else data.ctypes.data_as(ctypes.c_void_p)
        n = len(data) if buf is not None else data.nbytes
        self._check(self.lib.zarc_gpu_memcpy_h2d(self.h, ctypes.c_void_p(dptr), src, n))

    def d2h(self, dptr, nbytes):
        out = np.empty(nbytes, dtype=np.uint8)
        self._check(self.lib.zarc_gpu_memcpy_d2h(self.h, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(dptr), nbytes))
        return out

    # ---- device-resident batch calls ----
    def corpus_fill(self, dptr, off, length, first_index=0, kind=-1):
        off, poff = _u64(off)
        length, plen = _u64(length)
        self._check(self.lib.zarc_gpu_corpus_fill_device(self.h, len(off), ctypes.c_void_p(dptr), poff, plen, first_index, kind))

    def blake3_device(self, dptr, off, length):
        off, poff = _u64(off)
        length, plen = _u64(length)
        dig = np.zeros((len(off), 32), dtype=np.uint8)
        self._check(self.lib.zarc_gpu_blake3_batch_device(self.h, len(off), ctypes.c_void_p(dptr), poff, plen, dig.ctypes.data_as(ctypes.c_void_p)))
        return dig

    def xxh64_device(self, dptr, off, length):
        off, poff = _u64(off)
        length, plen = _u64(length)
        out = np.zeros(len(off), dtype=np.uint64)
        self._check(self.lib.zarc_gpu_xxh64_batch_device(self.h, len(off), ctypes.c_void_p(dptr), poff, plen,
                                                         out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))))
        return out

    def pack_device(self, d_src, off, length, d_dst, dst_cap):
        off, poff = _u64(off)
        length, plen = _u64(length)
        n = len(off)
        dst_off = np.zeros(n, dtype=np.uint64)
        dst_len = np.zeros(n, dtype=np.uint64)
        dig = np.zeros((n, 32), dtype=np.uint8)
        status = np.zeros(n, dtype=np.int32)
        self._check(self.lib.zarc_gpu_pack_batch_device(
            self.h, n, ctypes.c_void_p(d_src), poff, plen, ctypes.c_void_p(d_dst), dst_cap,
            dst_off.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), dst_len.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
            dig.ctypes.data_as(ctypes.c_void_p), status.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
        return dst_off, dst_len, dig, status

    def unpack_device(self, d_frames, frame_off, frame_len, d_dst, dst_off, raw_len, expect=None):
        frame_off, pfo = _u64(frame_off)
        frame_len, pfl = _u64(frame_len)
        dst_off, pdo = _u64(dst_off)
        raw_len, prl = _u64(raw_len)
        n = len(frame_off)
        dig = np.zeros((n, 32), dtype=np.uint8)
        status = np.zeros(n, dtype=np.int32)
        exp = None
        if expect is not None:
            exp = np.ascontiguousarray(expect, dtype=np.uint8)
        self._check(self.lib.zarc_gpu_unpack_batch_device(
            self.h, n, ctypes.c_void_p(d_frames), pfo, pfl, ctypes.c_void_p(d_dst), pdo, prl,
            exp.ctypes.data_as(ctypes.c_void_p) if exp is not None else None,
            dig.ctypes.data_as(ctypes.c_void_p), status.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
        return dig, status

    # ---- host-memory batch calls (the shape of the reference's API: slices in, bytes out) ----
    def blake3(self, entries):
        n = len(entries)
        bufs = [bytes(e) for e in entries]
        ptrs = (ctypes.c_void_p * n)(*[ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p) for b in bufs])
        lens = (ctypes.c_size_t * n)(*[len(b) for b in bufs])
        dig = np.zeros((n, 32), dtype=np.uint8)
        self._check(self.lib.zarc_gpu_blake3_batch(self.h, n, ptrs, lens, dig.ctypes.data_as(ctypes.c_void_p)))
        return [bytes(d) for d in dig]

    def pack(self, entries):
        """-> list of (frame_bytes, digest).  Mirrors Encoder::add_data_frame for a batch of entries."""
        n = len(entries)
        bufs = [bytes(e) for e in entries]
        ptrs = (ctypes.c_void_p * n)(*[ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p) for b in bufs])
        lens = (ctypes.c_size_t * n)(*[len(b) for b in bufs])
        cap = sum(self.bound(len(b)) for b in bufs)
        dst = np.zeros(max(cap, 1), dtype=np.uint8)
        dst_off = (ctypes.c_size_t * n)()
        dst_len = (ctypes.c_size_t * n)()
        dig = np.zeros((n, 32), dtype=np.uint8)
        status = (ctypes.c_int * n)()
        self._check(self.lib.zarc_gpu_pack_batch(self.h, n, ptrs, lens, dst.ctypes.data_as(ctypes.c_void_p), cap, dst_off, dst_len,
                                                 dig.ctypes.data_as(ctypes.c_void_p), status))
        return [(bytes(dst[dst_off[i]:dst_off[i] + dst_len[i]]), bytes(dig[i])) for i in range(n)]

    def pack_dedup(self, entries, seen):
        """Hash-first pack (Encoder::add_data_frame: hash, look the digest up, compress only new content; content_frame.rs:26-33).
        `seen` is a set of digests (bytes) the caller has frames for; it is updated with what this call compresses.
        -> list of (frame_bytes or None for a skipped duplicate, digest, status)."""
        n = len(entries)
        bufs = [bytes(e) for e in entries]
        ptrs = (ctypes.c_void_p * n)(*[ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p) for b in bufs])
        lens = (ctypes.c_size_t * n)(*[len(b) for b in bufs])
        cap = sum(self.bound(len(b)) for b in bufs)
        dst = np.zeros(max(cap, 1), dtype=np.uint8)
        dst_off, dst_len = (ctypes.c_size_t * n)(), (ctypes.c_size_t * n)()
        dig = np.zeros((n, 32), dtype=np.uint8)
        status = (ctypes.c_int * n)()
        calls = []

        def known(_ctx, d, i):
            key = bytes(d[:32])
            calls.append(i)
            if key in seen:
                return 1
            seen.add(key)
            return 0
        cb = _lib.KNOWN_FN(known)
        self._check(self.lib.zarc_gpu_pack_batch_dedup(self.h, n, ptrs, lens, dst.ctypes.data_as(ctypes.c_void_p), cap, dst_off, dst_len,
                                                       dig.ctypes.data_as(ctypes.c_void_p), status, cb, None))
        assert calls == list(range(n))       # once per entry, in index order
        return [(bytes(dst[dst_off[i]:dst_off[i] + dst_len[i]]) if status[i] != _lib.FRAME_DUPLICATE else None, bytes(dig[i]), int(status[i]))
                for i in range(n)]

    def pack_device_dedup(self, d_src, off, length, d_dst, dst_cap, seen):
        off, poff = _u64(off)
        length, plen = _u64(length)
        n = len(off)
        dst_off, dst_len = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
        dig = np.zeros((n, 32), dtype=np.uint8)
        status = np.zeros(n, dtype=np.int32)

        def known(_ctx, d, i):
            key = bytes(d[:32])
            if key in seen:
                return 1
            seen.add(key)
            return 0
        cb = _lib.KNOWN_FN(known)
        self._check(self.lib.zarc_gpu_pack_batch_device_dedup(
            self.h, n, ctypes.c_void_p(d_src), poff, plen, ctypes.c_void_p(d_dst), dst_cap,
            dst_off.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), dst_len.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
            dig.ctypes.data_as(ctypes.c_void_p), status.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), cb, None))
        return dst_off, dst_len, dig, status

    def unpack(self, frames, raw_lens, expect=None):
        """-> list of (bytes, digest, status).  Mirrors read_content_frame + FrameIterator::verify for a batch."""
        n = len(frames)
        bufs = [bytes(f) for f in frames]
        ptrs = (ctypes.c_void_p * n)(*[ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p) for b in bufs])
        lens = (ctypes.c_size_t * n)(*[len(b) for b in bufs])
        rl = (ctypes.c_size_t * n)(*[int(r) for r in raw_lens])
        outs = [np.zeros(max(int(r), 1), dtype=np.uint8) for r in raw_lens]
        optrs = (ctypes.c_void_p * n)(*[o.ctypes.data for o in outs])
        dig = np.zeros((n, 32), dtype=np.uint8)
        status = (ctypes.c_int * n)()
        exp = None
        if expect is not None:
            exp = np.ascontiguousarray(np.frombuffer(b"".join(expect), dtype=np.uint8))
        self._check(self.lib.zarc_gpu_unpack_batch(self.h, n, ptrs, lens, rl, optrs,
                                                   exp.ctypes.data_as(ctypes.c_void_p) if exp is not None else None,
                                                   dig.ctypes.data_as(ctypes.c_void_p), status))
        return [(bytes(outs[i][:int(raw_lens[i])]), bytes(dig[i]), int(status[i])) for i in range(n)]
