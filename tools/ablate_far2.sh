#!/bin/bash
# timing-only: 12288 = lookups only; +16384 plain (L1-cached) loads; +32768 lookups at even positions only
for d in 12288 28672 45056 61440 32768; do echo -n "DBG=$d "; ZARC_GPU_LIB=$PWD/zarc_amd/libzarc_gpu_diag.so ZARC_GPU_DBG=$d timeout -k 10 300 python bench.py --entries ${N:-10000} --steps 2 --warmup 1 --no-cpu-baseline --kind ${KIND:--1} 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); print(d['kernel_ms'])" || exit 1; done
