#!/bin/bash
# unpack A/B: tools/ab_unpack.sh lib...   (N entries of SZ bytes; default configs[1])
for lib in "$@"; do echo -n "$lib "; ZARC_GPU_LIB=$PWD/$lib timeout -k 10 300 python bench.py --entries ${N:-10000} --size ${SZ:-1048576} --steps 3 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null > /tmp/ab_unpack_$$.json; python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['unpack_gibs'], d['unpack_ms_per_step'], d['roundtrip_bit_exact'], d['unpack_kernel_ms'])" /tmp/ab_unpack_$$.json; done
