#!/bin/bash
# round 4: the decoder's frame order with / without the density key (diagnostic build)
cd $GRAFT_REPO_ROOT
P='import sys,json; l=[x for x in sys.stdin if x.startswith("{")]; d=json.loads(l[-1]); print(d["unpack_gibs"], d["roundtrip_bit_exact"], d["unpack_kernel_ms"])'
for r in 1 2; do for v in 0 1; do
  for c in "" "--config c5 --gib 24" "--config small" "--config c4 --gib 8"; do echo -n "density=$v [$c] "; ZARC_GPU_DEC_DENSITY=$v ZARC_GPU_LIB=$PWD/zarc_amd/libzarc_gpu_diag.so timeout -k 10 500 python bench.py $c --steps 3 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "$P"; done
done; done
