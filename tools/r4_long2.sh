#!/bin/bash
# round 4: libzstd-made frames and the mixed configurations on the product library (two runs)
cd $GRAFT_REPO_ROOT
P='import sys,json; l=[x for x in sys.stdin if x.startswith("{")]; d=json.loads(l[-1]); print(d["unpack_gibs"], d["roundtrip_bit_exact"], d["unpack_kernel_ms"])'
for r in 1 2; do
  timeout -k 10 400 python tools/libzstd_frames_rate.py 2048 3 9 19 2>&1 | tail -4
  for c in "--config small" "--config c5 --gib 24"; do echo -n "[$c] "; timeout -k 10 500 python bench.py $c --steps 3 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "$P"; done
done
