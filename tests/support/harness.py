"""Test-side helpers: ctypes bindings for the oracle, the corpus generator and system libzstd builds.

Test infrastructure only -- nothing under zarc_amd/ imports this.
"""
import ctypes
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _build_oracle():
    so = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
    srcs = glob.glob(os.path.join(ROOT, "oracle", "*.c")) + glob.glob(os.path.join(ROOT, "oracle", "*.h"))
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return so


def _build_corpus():
    d = os.path.join(ROOT, "tests", "support", "_build")
    os.makedirs(d, exist_ok=True)
    so = os.path.join(d, "libcorpus.so")
    srcs = [os.path.join(ROOT, "tests", "support", "corpus_shim.c"), os.path.join(ROOT, "zarc_amd", "csrc", "corpus.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", so, srcs[0]])
    return so


def _build_cpu_baseline():
    d = os.path.join(ROOT, "tests", "support", "_build")
    os.makedirs(d, exist_ok=True)
    so = os.path.join(d, "libcpubaseline.so")
    srcs = [os.path.join(ROOT, "tests", "support", "cpu_baseline.c"), os.path.join(ROOT, "oracle", "blake3_ref.c"),
            os.path.join(ROOT, "oracle", "oracle.h"), os.path.join(ROOT, "zarc_amd", "csrc", "corpus.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-pthread", "-o", so, srcs[0], srcs[1], "-ldl"])
    return so


def cpu_baseline(libzstd_path, level, threads, n, entry_bytes, first_index=0, kind=-1):
    """The reference's CPU path (one CCtx + session reset per entry, decompressStream in 131 075 / 131 072 byte steps, BLAKE3 on
    both sides) on `threads` host threads: tests/support/cpu_baseline.c.  Returns a dict or raises."""
    lib = ctypes.CDLL(_build_cpu_baseline())
    lib.cpu_baseline_run.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_int,
                                     ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64),
                                     ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    tp, tu, cb, hp, hu = ctypes.c_double(), ctypes.c_double(), ctypes.c_uint64(), ctypes.c_double(), ctypes.c_double()
    info = ctypes.create_string_buffer(400)
    rc = lib.cpu_baseline_run(libzstd_path.encode(), level, threads, n, entry_bytes, first_index, kind, ctypes.byref(tp), ctypes.byref(tu),
                              ctypes.byref(cb), info, len(info), ctypes.byref(hp), ctypes.byref(hu))
    if rc != 0:
        raise RuntimeError("cpu_baseline_run failed: %d" % rc)
    return {"pack_seconds": tp.value, "unpack_seconds": tu.value, "compressed_bytes": cb.value, "info": info.value.decode(),
            "bytes": n * entry_bytes, "threads": threads, "pack_hash_seconds": hp.value, "unpack_hash_seconds": hu.value}


class ZgeParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in (
        "level", "checksum", "window_log", "long_log", "short_log", "short_bytes", "tile", "sub", "cap",
        "min_match", "min_rep", "rep_search", "back_cap", "lazy", "lazy_delta", "lit_cost", "match_cost",
        "rep_cost", "short_window_log", "rep_back", "tag_bits", "seg_log", "far_log", "far_ways", "far_step_log", "far_res_log", "far_short", "far_skip", "far_back", "near16", "far_cdc_log", "far_min_frame", "rep_pass", "lazy2_delta", "far_cap", "cont_cap", "ext_cap", "live_reps", "seq_repeat")]


class ZgeStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("seqs", "rep_seqs", "match_bytes", "lit_bytes", "lit_section", "seq_section")] + \
               [(n, ctypes.c_uint32) for n in ("blk_raw", "blk_rle", "blk_comp", "lit_raw", "lit_rle", "lit_huf")] + \
               [("seq_mode", ctypes.c_uint32 * 4)]


class Oracle:
    def __init__(self):
        self.lib = o = ctypes.CDLL(_build_oracle())
        o.oracle_xxh64.restype = ctypes.c_uint64
        o.oracle_xxh64.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint64]
        o.oracle_blake3.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
        o.oracle_zstd_decode_frame.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                                               ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
        o.zge_bound.restype = ctypes.c_size_t
        o.zge_bound.argtypes = [ctypes.c_size_t]
        o.zge_default_params.argtypes = [ctypes.POINTER(ZgeParams), ctypes.c_int]
        o.zge_encode_frame.argtypes = [ctypes.POINTER(ZgeParams), ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p,
                                       ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ZgeStats)]

    def blake3(self, data):
        out = ctypes.create_string_buffer(32)
        self.lib.oracle_blake3(bytes(data), len(data), out)
        return out.raw

    def xxh64(self, data, seed=0):
        return self.lib.oracle_xxh64(bytes(data), len(data), seed)

    def zstd_decode(self, frame, cap):
        dst = ctypes.create_string_buffer(cap + 1)
        dl, cons = ctypes.c_size_t(), ctypes.c_size_t()
        rc = self.lib.oracle_zstd_decode_frame(bytes(frame), len(frame), dst, cap, ctypes.byref(dl), ctypes.byref(cons))
        return rc, dst.raw[:dl.value] if rc == 0 else b"", cons.value

    def params(self, level=3, **kw):
        p = ZgeParams()
        self.lib.zge_default_params(ctypes.byref(p), level)
        for k, v in kw.items():
            setattr(p, k, v)
        return p

    def zge_encode(self, data, params=None, stats=False):
        p = params or self.params()
        cap = self.lib.zge_bound(len(data))
        dst = ctypes.create_string_buffer(cap)
        ol = ctypes.c_size_t()
        st = ZgeStats()
        rc = self.lib.zge_encode_frame(ctypes.byref(p), bytes(data), len(data), dst, cap, ctypes.byref(ol), ctypes.byref(st))
        assert rc == 0, rc
        return (dst.raw[:ol.value], st) if stats else dst.raw[:ol.value]


class Corpus:
    def __init__(self):
        self.lib = ctypes.CDLL(_build_corpus())
        self.lib.corpus_entry.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_int]

    def entry(self, index, n, kind=-1):
        b = ctypes.create_string_buffer(n)
        self.lib.corpus_entry(b, n, index, kind)
        return b.raw


LIBZSTD_CANDIDATES = ["/usr/lib/x86_64-linux-gnu/libzstd.so.1", "/opt/conda/lib/libzstd.so.1"] + \
    sorted(glob.glob("/usr/local/lib/python3.10/dist-packages/pillow.libs/libzstd*"))


class LibZstd:
    """A real libzstd build loaded with ctypes (the reference's own codec, at a different pin)."""

    def __init__(self, path):
        self.path = path
        self.z = z = ctypes.CDLL(path)
        z.ZSTD_versionString.restype = ctypes.c_char_p
        self.version = z.ZSTD_versionString().decode()
        z.ZSTD_createCCtx.restype = ctypes.c_void_p
        z.ZSTD_freeCCtx.argtypes = [ctypes.c_void_p]
        z.ZSTD_compress2.restype = ctypes.c_size_t
        z.ZSTD_compress2.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
        z.ZSTD_CCtx_setParameter.restype = ctypes.c_size_t
        z.ZSTD_CCtx_setParameter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        z.ZSTD_decompress.restype = ctypes.c_size_t
        z.ZSTD_decompress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
        z.ZSTD_isError.argtypes = [ctypes.c_size_t]
        z.ZSTD_getErrorName.restype = ctypes.c_char_p
        z.ZSTD_getErrorName.argtypes = [ctypes.c_size_t]

    def compress(self, data, level=3, checksum=1):
        z = self.z
        c = z.ZSTD_createCCtx()
        z.ZSTD_CCtx_setParameter(c, 100, level)
        z.ZSTD_CCtx_setParameter(c, 201, checksum)
        cap = len(data) + max(1024, len(data) // 10)  # crates/zarc/src/encode/lowlevel_frames.rs:21
        dst = ctypes.create_string_buffer(cap)
        n = z.ZSTD_compress2(c, dst, cap, bytes(data), len(data))
        z.ZSTD_freeCCtx(c)
        assert not z.ZSTD_isError(n), z.ZSTD_getErrorName(n)
        return dst.raw[:n]

    def decompress(self, frame, cap):
        dst = ctypes.create_string_buffer(cap + 1)
        n = self.z.ZSTD_decompress(dst, cap, bytes(frame), len(frame))
        if self.z.ZSTD_isError(n):
            return None, self.z.ZSTD_getErrorName(n).decode()
        return dst.raw[:n], None


def libzstds():
    out = []
    for p in LIBZSTD_CANDIDATES:
        if os.path.exists(p):
            try:
                out.append(LibZstd(p))
            except OSError:
                pass
    return out
