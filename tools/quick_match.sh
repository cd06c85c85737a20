#!/bin/bash
# quick look at the match finder on the GPU box: bench at N entries (all kinds, then each kind), then the stage profile
N=${N:-4096}
for k in -1 0 1 2 3; do echo -n "kind=$k "; timeout -k 10 300 python bench.py --entries $N --steps 2 --warmup 1 --no-cpu-baseline --kind $k 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); print(d['value'], d['unpack_gibs'], d['ratio'], d['kernel_ms'], d['unpack_kernel_ms'])" || exit 1; done
ZARC_GPU_DBG=1024 timeout -k 10 300 python bench.py --entries $N --steps 1 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep -v "^$" | tail -4
