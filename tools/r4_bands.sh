#!/bin/bash
# round 4: decoder frame order, 4 KiB size steps (1) against eighth-of-an-octave size bands (2) (diagnostic build zarc_amd/ab/bands.so)
cd $GRAFT_REPO_ROOT
P='import sys,json; l=[x for x in sys.stdin if x.startswith("{")]; d=json.loads(l[-1]); print(d["unpack_gibs"], d["roundtrip_bit_exact"], d["unpack_kernel_ms"])'
for r in 1 2; do for v in 1 2; do
  for c in "--config c5 --gib 24" "--config small" ""; do echo -n "order=$v [$c] "; ZARC_GPU_DEC_DENSITY=$v ZARC_GPU_LIB=$PWD/zarc_amd/ab/bands.so timeout -k 10 500 python bench.py $c --steps 3 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "$P"; done
done; done
