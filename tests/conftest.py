import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "support"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import harness  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    return harness.Oracle()


@pytest.fixture(scope="session")
def corpus():
    return harness.Corpus()


@pytest.fixture(scope="session")
def libzstds():
    return harness.libzstds()


@pytest.fixture(scope="session")
def golden_frames():
    d = os.path.join(ROOT, "tests", "golden", "zstd_frames")
    m = json.load(open(os.path.join(d, "manifest.json")))
    return d, m


@pytest.fixture(scope="session")
def emu_lib_path():
    """The kernels compiled against the HIP emulator (test infrastructure; see tests/emu/hip/hip_runtime.h)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    return os.path.join(ROOT, "tests", "emu", "_build", "libzarc_gpu_emu.so")


@pytest.fixture(scope="session")
def emu_engine(emu_lib_path):
    from zarc_amd import Engine, _lib
    e = Engine(0, emu_lib_path)
    e.set_parameter(_lib.P_CHECKSUM_FLAG, 1)
    return e


@pytest.fixture(scope="session")
def engine():
    """The product library on a real GPU.  No fallback: fails if libzarc_gpu.so or the device is missing."""
    from zarc_amd import Engine, _lib
    e = Engine(0)
    e.set_parameter(_lib.P_CHECKSUM_FLAG, 1)  # crates/zarc-cli/src/pack.rs:227
    return e
