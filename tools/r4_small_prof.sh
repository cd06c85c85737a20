#!/bin/bash
# round 4: stage clocks of the entropy stage and the match finder on 4 KiB entries (diagnostic build)
cd $GRAFT_REPO_ROOT
ZARC_GPU_LIB=$PWD/zarc_amd/libzarc_gpu_diag.so ZARC_GPU_DBG=1024 timeout -k 10 300 python bench.py --entries 524288 --size 4096 --steps 1 --warmup 1 --no-cpu-baseline --no-host-path 2>&1 >/dev/null | grep "stage ticks" | tail -2
ZARC_GPU_LIB=$PWD/zarc_amd/libzarc_gpu_diag.so ZARC_GPU_DBG=512 timeout -k 10 300 python bench.py --entries 524288 --size 4096 --steps 1 --warmup 1 --no-cpu-baseline --no-host-path 2>&1 >/dev/null | grep "host phases" | tail -2
