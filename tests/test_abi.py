"""CPU: the product library loads, exports every symbol include/zarc_gpu.h declares, and refuses to run
without a GPU (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def product_lib():
    so = os.path.join(ROOT, "zarc_amd", "libzarc_gpu.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "zarc_amd", "csrc")])
    return so


def _declared():
    text = open(os.path.join(ROOT, "include", "zarc_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zarc_gpu_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from zarc_amd import _lib
    assert sorted(_lib.EXPORTS) == _declared()


def test_library_exports_every_declared_symbol(product_lib):
    lib = ctypes.CDLL(product_lib)
    for name in _declared():
        assert hasattr(lib, name), name
    lib.zarc_gpu_abi_version.restype = ctypes.c_int
    assert lib.zarc_gpu_abi_version() == 2


def test_bound_and_error_names(product_lib):
    from zarc_amd import _lib
    lib = _lib.load(product_lib)
    assert lib.zarc_gpu_bound(0) == 32 and lib.zarc_gpu_bound(65536) == 65568 and lib.zarc_gpu_bound(131073) == 131104   # three 64 KiB blocks
    assert lib.zarc_gpu_bound(131073) >= 131073 + 6 + 18
    assert lib.zarc_gpu_frame_status_name(1) == b"Data corruption detected"
    assert lib.zarc_gpu_frame_status_name(2) == b"Restored data doesn't match checksum"
    assert lib.zarc_gpu_error_name(-5) == b"Destination buffer is too small"


def test_no_cpu_fallback_without_device(product_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from zarc_amd import Engine, ZarcGpuError
    with pytest.raises(ZarcGpuError):
        Engine(0, product_lib)


def test_product_never_touches_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "zarc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in text and "oracle_" not in text and "hipemu" not in text.replace("ZARC_HIPEMU", "").replace("hipemu_", ""), f
