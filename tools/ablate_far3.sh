#!/bin/bash
# timing-only: 16384 = waves 0/1 make no next-tile far lookups; 32768 = nobody does (entries stay empty)
for d in 0 16384 32768; do echo -n "DBG=$d "; ZARC_GPU_LIB=$PWD/zarc_amd/libzarc_gpu_diag.so ZARC_GPU_DBG=$d timeout -k 10 300 python bench.py --entries ${N:-10000} --steps 2 --warmup 1 --no-cpu-baseline --no-host-path --kind ${KIND:--1} 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); print(d['kernel_ms'])" || exit 1; done
