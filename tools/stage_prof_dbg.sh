#!/bin/bash
# stage clocks of the match finder with extra ZARC_GPU_DBG ablation bits: tools/stage_prof_dbg.sh lib.so kind bits...
LIB=$1; K=$2; shift 2
for b in "$@"; do echo -n "dbg=$b kind=$K "; ZARC_GPU_LIB=$PWD/$LIB ZARC_GPU_DBG=$((1024 + b)) timeout -k 10 300 python bench.py --entries ${N:-4096} --steps 1 --warmup 1 --no-cpu-baseline --no-host-path --kind $K 2>&1 >/dev/null | grep "zge_match stage ticks" | tail -1; done
