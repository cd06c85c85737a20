#!/bin/bash
# decoder A/B (diagnostic builds): tools/r4_ab_dec2.sh lib...   -- stage-2 kernels alone (side=0) and side by side, two runs each
cd $GRAFT_REPO_ROOT
P='import sys,json; l=[x for x in sys.stdin if x.startswith("{")]; d=json.loads(l[-1]); print(d["unpack_gibs"], d["roundtrip_bit_exact"], d["unpack_kernel_ms"])'
for r in 1 2; do for lib in "$@"; do for side in 0 1; do
  echo -n "$lib side=$side "; ZARC_GPU_DEC_SIDE=$side ZARC_GPU_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "$P"
done; done; done
