#!/bin/bash
# timing-only ablations of the far-table path (diagnostic build): 2048 no lookups, 4096 candidates ignored, 8192 no inserts
for d in 0 14336 4096 8192 12288 10240; do echo -n "DBG=$d "; ZARC_GPU_LIB=$PWD/zarc_amd/libzarc_gpu_diag.so ZARC_GPU_DBG=$d timeout -k 10 300 python bench.py --entries ${N:-10000} --steps 2 --warmup 1 --no-cpu-baseline --kind ${KIND:--1} 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); print(d['kernel_ms'])" || exit 1; done
