#!/bin/bash
# round 4: entropy passes pipelined in chunks over two streams (ZARC_GPU_ENT_CHUNKS, diagnostic build) against one chunk
cd $GRAFT_REPO_ROOT
for c in 1 2 4 8; do echo -n "chunks=$c "; ZARC_GPU_ENT_CHUNKS=$c ZARC_GPU_LIB=$PWD/zarc_amd/libzarc_gpu_diag.so timeout -k 10 300 python bench.py --no-cpu-baseline --no-host-path --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pack', d['value'], d['ms_per_step'], 'unpack', d['unpack_gibs'], d['roundtrip_bit_exact'], d['kernel_ms'])"; done
echo -n "product "; timeout -k 10 300 python bench.py --no-cpu-baseline --no-host-path --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pack', d['value'], d['ms_per_step'], 'unpack', d['unpack_gibs'], d['roundtrip_bit_exact'], d['kernel_ms'], d['unpack_kernel_ms'])"
