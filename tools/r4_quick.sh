#!/bin/bash
# round 4: one line per configuration (small, 4 KiB entries, c5, configs[1]) on the product library
cd $GRAFT_REPO_ROOT
P='import sys,json; l=[x for x in sys.stdin if x.startswith("{")]; d=json.loads(l[-1]); print(d["value"], d["unpack_gibs"], d["roundtrip_bit_exact"], d["unpack_kernel_ms"])'
for c in "--config small" "--config c5 --gib 24" ""; do echo -n "[$c] "; timeout -k 10 500 python bench.py $c --steps 3 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "$P"; done
bash tools/ab_size.sh $((2147483648 / 4096)) 4096 zarc_amd/libzarc_gpu.so
