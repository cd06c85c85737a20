"""ctypes binding of the C ABI in include/zarc_gpu.h (libzarc_gpu.so, built by zarc_amd/csrc/Makefile).

The library is the product: HIP kernels for gfx950 behind an extern "C" boundary.  There is no Python or
CPU implementation of the data path -- if the shared object is missing or no HIP device is usable this
module raises instead of degrading.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libzarc_gpu.so")

ABI_VERSION = 2  # include/zarc_gpu.h: ZARC_GPU_ABI_VERSION
DIGEST_LEN = 32
ALIGN = 16
PAD = 64

# call-level errors
OK, E_DEVICE, E_NOMEM, E_PARAM, E_UNSUPPORTED, E_DSTSIZE = 0, -1, -2, -3, -4, -5
# per-frame status
FRAME_OK, FRAME_CORRUPT, FRAME_CHECKSUM, FRAME_DIGEST, FRAME_DSTSIZE, FRAME_BAD_MAGIC, FRAME_UNSUPPORTED, FRAME_SRCSIZE, FRAME_DUPLICATE = range(9)
# parameter ids (ZSTD_cParameter values, what zstd_safe::CParameter maps to)
P_COMPRESSION_LEVEL, P_WINDOW_LOG, P_HASH_LOG, P_CHAIN_LOG, P_SEARCH_LOG, P_MIN_MATCH, P_TARGET_LENGTH, P_STRATEGY = 100, 101, 102, 103, 104, 105, 106, 107
P_ENABLE_LDM, P_LDM_HASH_LOG, P_LDM_MIN_MATCH, P_LDM_BUCKET_SIZE_LOG, P_LDM_HASH_RATE_LOG = 160, 161, 162, 163, 164
P_CONTENT_SIZE_FLAG, P_CHECKSUM_FLAG, P_DICT_ID_FLAG = 200, 201, 202
# engine tuning (batching only; frames are identical for every value)
PX_SCRATCH_MB, PX_STAGE_CHUNK, PX_STAGE_THREAD, PX_COPY_THREADS, PX_DEC_GROUPS, PX_ZERO_COPY = 9001, 9002, 9003, 9004, 9005, 9006
# timers
T_BLAKE3, T_XXH64, T_MATCH, T_ENTROPY, T_ASSEMBLE, T_DECODE, T_TOTAL, T_DEC_SEQS, T_DEC_LITS, T_DEC_FRAMES = range(10)

EXPORTS = [
    "zarc_gpu_abi_version", "zarc_gpu_level_finder", "zarc_gpu_parameter_advisory", "zarc_gpu_device_count", "zarc_gpu_create", "zarc_gpu_destroy", "zarc_gpu_set_parameter", "zarc_gpu_get_params",
    "zarc_gpu_enable_compression", "zarc_gpu_bound", "zarc_gpu_error_name", "zarc_gpu_frame_status_name", "zarc_gpu_last_error",
    "zarc_gpu_pack_batch", "zarc_gpu_pack_batch_device", "zarc_gpu_pack_batch_dedup", "zarc_gpu_pack_batch_device_dedup", "zarc_gpu_unpack_batch", "zarc_gpu_unpack_batch_device",
    "zarc_gpu_blake3_batch", "zarc_gpu_blake3_batch_device", "zarc_gpu_xxh64_batch_device", "zarc_gpu_last_kernel_ms",
    "zarc_gpu_corpus_fill_device", "zarc_gpu_device_malloc", "zarc_gpu_device_free", "zarc_gpu_memcpy_h2d", "zarc_gpu_memcpy_d2h",
]


# int known(void *ctx, const uint8_t digest[32], size_t index): nonzero = the caller has this content already (hash-first dedup)
KNOWN_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.c_size_t)


class Params(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("level", "checksum_flag", "content_size_flag", "window_log", "hash_log", "chain_log",
                                            "search_log", "min_match", "target_length", "strategy", "compress")]


class ZarcGpuError(RuntimeError):
    def __init__(self, code, name, detail=""):
        super().__init__("%s (%d)%s" % (name, code, (": " + detail) if detail else ""))
        self.code = code


def load(path=None):
    """Load the shared library and declare every entry point.  Raises OSError if it is not built."""
    path = path or os.environ.get("ZARC_GPU_LIB") or DEFAULT_LIB
    if not os.path.exists(path):
        raise OSError("libzarc_gpu.so not found at %s -- build it with `make -C zarc_amd/csrc` "
                      "(or __graft_entry__.build()); there is no CPU fallback" % path)
    lib = ctypes.CDLL(path)
    c = ctypes
    vp, sz, u64p, u8p, ip = c.c_void_p, c.c_size_t, c.POINTER(c.c_uint64), c.POINTER(c.c_uint8), c.POINTER(c.c_int)
    szp, vpp = c.POINTER(c.c_size_t), c.POINTER(c.c_void_p)
    lib.zarc_gpu_abi_version.restype = c.c_int
    got = lib.zarc_gpu_abi_version()
    if got != ABI_VERSION:  # checked BEFORE the newer symbols are resolved: a stale library fails here, by version, not at a symbol lookup
        raise OSError("%s has ABI version %d, this binding needs %d -- rebuild it (make -C zarc_amd/csrc)" % (path, got, ABI_VERSION))
    lib.zarc_gpu_device_count.restype = c.c_int
    lib.zarc_gpu_level_finder.argtypes = [c.c_int]
    lib.zarc_gpu_parameter_advisory.argtypes = [c.c_int]
    lib.zarc_gpu_create.argtypes = [c.POINTER(vp), c.c_int]
    lib.zarc_gpu_destroy.argtypes = [vp]
    lib.zarc_gpu_destroy.restype = None
    lib.zarc_gpu_set_parameter.argtypes = [vp, c.c_int, c.c_int]
    lib.zarc_gpu_get_params.argtypes = [vp, c.POINTER(Params)]
    lib.zarc_gpu_get_params.restype = None
    lib.zarc_gpu_enable_compression.argtypes = [vp, c.c_int]
    lib.zarc_gpu_enable_compression.restype = None
    lib.zarc_gpu_bound.argtypes = [sz]
    lib.zarc_gpu_bound.restype = sz
    for f in (lib.zarc_gpu_error_name, lib.zarc_gpu_frame_status_name):
        f.argtypes = [c.c_int]
        f.restype = c.c_char_p
    lib.zarc_gpu_last_error.argtypes = [vp]
    lib.zarc_gpu_last_error.restype = c.c_char_p
    lib.zarc_gpu_pack_batch.argtypes = [vp, sz, vpp, szp, vp, sz, szp, szp, vp, ip]
    lib.zarc_gpu_pack_batch_device.argtypes = [vp, sz, vp, u64p, u64p, vp, sz, u64p, u64p, vp, ip]
    lib.zarc_gpu_pack_batch_dedup.argtypes = [vp, sz, vpp, szp, vp, sz, szp, szp, vp, ip, KNOWN_FN, vp]
    lib.zarc_gpu_pack_batch_device_dedup.argtypes = [vp, sz, vp, u64p, u64p, vp, sz, u64p, u64p, vp, ip, KNOWN_FN, vp]
    lib.zarc_gpu_unpack_batch.argtypes = [vp, sz, vpp, szp, szp, vpp, vp, vp, ip]
    lib.zarc_gpu_unpack_batch_device.argtypes = [vp, sz, vp, u64p, u64p, vp, u64p, u64p, vp, vp, ip]
    lib.zarc_gpu_blake3_batch.argtypes = [vp, sz, vpp, szp, vp]
    lib.zarc_gpu_blake3_batch_device.argtypes = [vp, sz, vp, u64p, u64p, vp]
    lib.zarc_gpu_xxh64_batch_device.argtypes = [vp, sz, vp, u64p, u64p, u64p]
    lib.zarc_gpu_last_kernel_ms.argtypes = [vp, c.c_int]
    lib.zarc_gpu_last_kernel_ms.restype = c.c_float
    lib.zarc_gpu_corpus_fill_device.argtypes = [vp, sz, vp, u64p, u64p, c.c_uint64, c.c_int]
    lib.zarc_gpu_device_malloc.argtypes = [vp, c.POINTER(vp), sz]
    lib.zarc_gpu_device_free.argtypes = [vp, vp]
    lib.zarc_gpu_memcpy_h2d.argtypes = [vp, vp, vp, sz]
    lib.zarc_gpu_memcpy_d2h.argtypes = [vp, vp, vp, sz]
    return lib
