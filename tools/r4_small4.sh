#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "unpack or roundtrip or golden or fuzz" 2>&1 | tail -2
for c in "--config small" "--entries 524288 --size 4096" "--entries 131072 --size 16384" ""; do echo -n "[$c] "; timeout -k 10 400 python bench.py $c --steps 3 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['unpack_gibs'], d['roundtrip_bit_exact'], d['unpack_kernel_ms'])"; done
