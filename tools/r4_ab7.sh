#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 1 0; do echo -n "digest_late=$v "; ZARC_GPU_DIGEST_LATE=$v ZARC_GPU_LIB=$PWD/zarc_amd/libzarc_gpu_diag.so timeout -k 10 300 python bench.py --no-cpu-baseline --no-host-path --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pack', d['value'], d['ms_per_step'], 'unpack', d['unpack_gibs'], d['roundtrip_bit_exact'], d['kernel_ms'])"; done
