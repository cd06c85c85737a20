#!/bin/bash
# (round 4: ZARC_GPU_SEQ_LDS_FRAC is gone -- the long-block rule of zarc_zdec_seqs_lds replaced it; kept for the record) decoder A/B on the GPU box: tools/ab_dec.sh lib...   with FRACS="0 0.5 1" (share of blocks on the LDS sequence kernel)
N=${N:-10000}
for lib in "$@"; do for f in ${FRACS:-0 1}; do echo -n "$lib lds_frac=$f "; ZARC_GPU_SEQ_LDS_FRAC=$f ZARC_GPU_LIB=$PWD/$lib timeout -k 10 300 python bench.py --entries $N --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); print(d['unpack_gibs'], d['roundtrip_bit_exact'], d['unpack_kernel_ms'])"; done; done
