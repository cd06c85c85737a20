#!/bin/bash
# pack / unpack A/B at one entry size: tools/ab_size.sh entries size lib...
E=$1; S=$2; shift 2
for lib in "$@"; do echo -n "$lib entries=$E size=$S "; ZARC_GPU_LIB=$PWD/$lib timeout -k 10 250 python bench.py --entries $E --size $S --steps 2 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null > /tmp/ab_size_$$.json; python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['value'], d['unpack_gibs'], d['roundtrip_bit_exact'], d['ratio'], d['kernel_ms'])" /tmp/ab_size_$$.json; done
