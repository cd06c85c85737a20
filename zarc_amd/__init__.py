"""zarc_amd -- MI355X (gfx950) content engine for the Zarc archive format.

Scope: the per-entry content pipeline of passcod/zarc (BLAKE3 digest + Zstandard frame encode/decode +
XXH64 checksum) as hand-written HIP kernels behind the C ABI in include/zarc_gpu.h.  See DESIGN.md.
"""
from ._lib import ZarcGpuError  # noqa: F401
from .engine import Engine  # noqa: F401
