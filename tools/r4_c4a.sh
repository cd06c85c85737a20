#!/bin/bash
# round 4, level 9 on the configs[3] shape: product rate, then stage clocks and timing ablations of the deep finder (diagnostic build)
cd $GRAFT_REPO_ROOT
GIB=8 bash tools/ab_c4.sh zarc_amd/libzarc_gpu.so
GIB=8 bash tools/stage_prof_c4.sh zarc_amd/libzarc_gpu_diag.so 0
for b in 2048 8192 10240 4096; do echo -n "timing dbg=$b: "; ZARC_GPU_LIB=$PWD/zarc_amd/libzarc_gpu_diag.so ZARC_GPU_DBG=$b timeout -k 10 300 python bench.py --config c4 --gib 8 --steps 1 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); print(d['value'], d['kernel_ms'])"; done
