#!/bin/bash
# A/B of engine builds on the GPU box: tools/ab.sh ab/base.so ab/v1.so ...   (N entries, KINDS list)
N=${N:-10000}
for lib in "$@"; do for k in ${KINDS:--1}; do echo -n "$lib kind=$k "; ZARC_GPU_LIB=$PWD/$lib timeout -k 10 300 python bench.py --entries $N --steps 2 --warmup 1 --no-cpu-baseline --kind $k 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); print(d['value'], d['unpack_gibs'], d['ratio'], d['kernel_ms'], d['unpack_kernel_ms'])" || exit 1; done; done
