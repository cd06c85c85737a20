#!/bin/bash
# stage clocks of the match finder / entropy coder (diagnostic build): tools/stage_prof.sh lib.so [kinds...]
LIB=$1; shift
for k in ${@:--1}; do echo "== $LIB kind=$k"; ZARC_GPU_LIB=$PWD/$LIB ZARC_GPU_DBG=1024 timeout -k 10 300 python bench.py --entries ${N:-4096} --steps 1 --warmup 1 --no-cpu-baseline --kind $k 2>&1 >/dev/null | grep "stage ticks" | tail -2; done
