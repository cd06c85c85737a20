#!/bin/bash
# (round 4: ZARC_GPU_SEQ_LDS_FRAC is gone -- the long-block rule of zarc_zdec_seqs_lds replaced it; kept for the record) round 4: the small-entry batch through the LDS-table sequence kernel (16 lanes per workgroup, every lane its own tables in LDS)
cd $GRAFT_REPO_ROOT
for f in 0 1; do echo -n "lds_frac=$f "; ZARC_GPU_SEQ_LDS_FRAC=$f ZARC_GPU_LIB=$PWD/zarc_amd/libzarc_gpu_diag.so timeout -k 10 400 python bench.py --config small --steps 2 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['unpack_gibs'], d['roundtrip_bit_exact'], d['unpack_kernel_ms'])"; done
for f in 0 1; do echo -n "4KiB lds_frac=$f "; ZARC_GPU_SEQ_LDS_FRAC=$f ZARC_GPU_LIB=$PWD/zarc_amd/libzarc_gpu_diag.so timeout -k 10 400 python bench.py --entries 524288 --size 4096 --steps 2 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['unpack_gibs'], d['roundtrip_bit_exact'], d['unpack_kernel_ms'])"; done
