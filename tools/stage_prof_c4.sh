#!/bin/bash
# stage clocks of the level-9 match finder on the configs[3] shape (diagnostic build): tools/stage_prof_c4.sh lib.so [dbg bits...]
LIB=$1; shift
for b in ${@:-0}; do echo -n "dbg=$b "; ZARC_GPU_LIB=$PWD/$LIB ZARC_GPU_DBG=$((1024 + b)) timeout -k 10 300 python bench.py --config c4 --gib ${GIB:-8} --steps 1 --warmup 1 --no-cpu-baseline --no-host-path 2>&1 >/dev/null | grep "zge_match stage ticks" | tail -1; done
