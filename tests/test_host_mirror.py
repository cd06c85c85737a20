"""The C++ host-side mirror of the reference's Encoder / read_content_frame API (zarc_amd/host/zarc_host.hpp):
dedup, call-order offsets from 12, config C1's frame shape, verify semantics.  CPU: linked against the emulated
build of the kernels; GPU: linked against the product library."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_mirror_emulated(emu_lib_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "host"])
    out = subprocess.check_output([os.path.join(ROOT, "tests", "emu", "_build", "host_mirror_test"), "9000"], timeout=600)
    assert b"host mirror OK" in out


@pytest.mark.gpu
def test_host_mirror_gpu():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "zarc_amd", "csrc"), "host"])
    out = subprocess.check_output([os.path.join(ROOT, "zarc_amd", "host_mirror_test"), "3000000"], timeout=600)
    assert b"host mirror OK" in out
