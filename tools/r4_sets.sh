#!/bin/bash
# round 4: sequence kernel -- table sets per workgroup x lanes per workgroup (diagnostic builds), stages alone and side by side
cd $GRAFT_REPO_ROOT
P='import sys,json; l=[x for x in sys.stdin if x.startswith("{")]; d=json.loads(l[-1]); print(d["unpack_gibs"], d["roundtrip_bit_exact"], d["unpack_kernel_ms"])'
for lib in zarc_amd/libzarc_gpu_diag.so zarc_amd/ab/sa.so zarc_amd/ab/sb.so; do for w in 32 64; do for side in 0 1; do
  echo -n "$lib width=$w side=$side "; ZARC_GPU_DEC_STATS=1 ZARC_GPU_SEQ_WIDTH=$w ZARC_GPU_DEC_SIDE=$side ZARC_GPU_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-path 2>gpurun_out/sets_err.txt | python -c "$P"; grep "turned down" gpurun_out/sets_err.txt | tail -1
done; done; done
