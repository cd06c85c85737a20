#!/bin/bash
# round 4: lean unpack sizing (device prefix sums, no piece list) against the host path (ZARC_GPU_DEC_LEAN=0, diagnostic build)
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "unpack or roundtrip or golden or fuzz" 2>&1 | tail -2
for lean in 1 0; do
echo -n "small lean=$lean "; ZARC_GPU_DEC_LEAN=$lean ZARC_GPU_LIB=$PWD/zarc_amd/libzarc_gpu_diag.so timeout -k 10 400 python bench.py --config small --steps 3 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['unpack_gibs'], d['roundtrip_bit_exact'], d['unpack_kernel_ms'])"
echo -n "4KiB lean=$lean "; ZARC_GPU_DEC_LEAN=$lean ZARC_GPU_LIB=$PWD/zarc_amd/libzarc_gpu_diag.so timeout -k 10 400 python bench.py --entries 524288 --size 4096 --steps 3 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['unpack_gibs'], d['roundtrip_bit_exact'], d['unpack_kernel_ms'])"
echo -n "c2 lean=$lean "; ZARC_GPU_DEC_LEAN=$lean ZARC_GPU_LIB=$PWD/zarc_amd/libzarc_gpu_diag.so timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['unpack_gibs'], d['roundtrip_bit_exact'], d['unpack_kernel_ms'])"
done
