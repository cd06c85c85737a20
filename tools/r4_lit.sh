#!/bin/bash
# round 4: literal kernel -- stages alone (diagnostic twin, one stream) and side by side, then the SQ counters
cd $GRAFT_REPO_ROOT
P='import sys,json; l=[x for x in sys.stdin if x.startswith("{")]; d=json.loads(l[-1]); print(d["value"], d["unpack_gibs"], d["roundtrip_bit_exact"], d["unpack_kernel_ms"])'
for side in 1 0; do echo -n "side=$side "; ZARC_GPU_DEC_SIDE=$side ZARC_GPU_LIB=$PWD/zarc_amd/libzarc_gpu_diag.so timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "$P"; done
echo -n "product "; timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "$P"
bash tools/pmc_decode.sh
