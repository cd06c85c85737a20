#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "pack_bit_exact or level_tiers or real_data" 2>&1 | tail -2
N=10000 bash tools/ab.sh zarc_amd/libzarc_gpu.so
echo -n "small: "; timeout -k 10 400 python bench.py --config small --steps 2 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['unpack_gibs'], d['ratio'], d['kernel_ms'], d['unpack_kernel_ms'])"
