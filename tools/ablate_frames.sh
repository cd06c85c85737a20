#!/bin/bash
# timing ablations of the decoder's frame pass on a chain-bound workload (few large frames), diagnostic build:
# tools/ablate_frames.sh [entries] [size] [kind]   -- ZARC_GPU_DBG_DEC bits: 1 no copies, 8 no near matches, 16 no flush, 32 no literal staging,
# 64 no far-match staging, 128 all near matches in order (outputs are invalid when set)
N=${1:-64}; SZ=${2:-16777216}; K=${3:-0}
for d in 0 1 8 16 32 64 96 128 120; do echo -n "dbg_dec=$d "; ZARC_GPU_DBG_DEC=$d ZARC_GPU_LIB=$PWD/zarc_amd/libzarc_gpu_diag.so timeout -k 10 300 python bench.py --entries $N --size $SZ --kind $K --steps 2 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); print(d['unpack_gibs'], d['unpack_kernel_ms'])" || exit 1; done
