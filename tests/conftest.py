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
def libzstd15(libzstds):
    """A libzstd 1.5.x build (the reference pins 1.5.5): the ratio yardstick.  The image carries one; a box without any libzstd
    fails the ratio and cross-decoding tests loudly instead of passing them vacuously."""
    assert libzstds, "no libzstd found on this box (looked at %s)" % (harness.LIBZSTD_CANDIDATES,)
    return next((z for z in libzstds if z.version.startswith("1.5")), libzstds[0])


@pytest.fixture(scope="session")
def real_items():
    import realdata
    items = realdata.items()
    missing = [k for k, v in items.items() if v is None]
    if missing:
        print("realdata: sources missing on this box, items skipped: %s" % missing)
    assert sum(v is not None for v in items.values()) >= 5, "hardly any real-data source on this box: %s" % missing
    return {k: v for k, v in items.items() if v is not None}


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
