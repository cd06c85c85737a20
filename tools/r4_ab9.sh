#!/bin/bash
# round 4, timing only: the fast finder with the lazy step (v3, --level 1), the level-3 finder without its extension round (v4)
cd $GRAFT_REPO_ROOT
run() { lib=$1; lvl=$2; echo -n "$lib level=$lvl "; ZARC_GPU_LIB=$PWD/$lib timeout -k 10 300 python bench.py --level $lvl --steps 2 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ratio'], d['kernel_ms'])"; }
run zarc_amd/libzarc_gpu.so 1; run zarc_amd/csrc/ab/v3.so 1; run zarc_amd/libzarc_gpu.so 3; run zarc_amd/csrc/ab/v4.so 3; run zarc_amd/libzarc_gpu.so 2
