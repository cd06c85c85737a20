#!/bin/bash
# round 4: 64 KiB encoder blocks -- parity on the GPU, then the configurations
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for c in "" "--config c5 --gib 24" "--config small" "--config c4 --gib 8" "--level -1"; do echo -n "[$c] "; timeout -k 10 500 python bench.py $c --steps 3 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pack', d['value'], 'unpack', d['unpack_gibs'], d['roundtrip_bit_exact'], 'ratio', d['ratio'], d['kernel_ms'], d['unpack_kernel_ms'])"; done
