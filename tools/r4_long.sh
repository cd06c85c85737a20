#!/bin/bash
# round 4: the long-block sequence kernel (every lane's own tables in LDS) on / off (diagnostic build): libzstd's frames, small, c5, configs[1]
cd $GRAFT_REPO_ROOT
P='import sys,json; l=[x for x in sys.stdin if x.startswith("{")]; d=json.loads(l[-1]); print(d["unpack_gibs"], d["roundtrip_bit_exact"], d["unpack_kernel_ms"])'
for v in 0 1; do
  echo "== ZARC_GPU_SEQ_LONG=$v"
  ZARC_GPU_SEQ_LONG=$v ZARC_GPU_LIB=$PWD/zarc_amd/libzarc_gpu_diag.so timeout -k 10 400 python tools/libzstd_frames_rate.py 2048 3 9 19 2>&1 | tail -8
  for c in "--config small" "--config c5 --gib 24" ""; do echo -n "[$c] "; ZARC_GPU_SEQ_LONG=$v ZARC_GPU_LIB=$PWD/zarc_amd/libzarc_gpu_diag.so timeout -k 10 500 python bench.py $c --steps 3 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "$P"; done
done
