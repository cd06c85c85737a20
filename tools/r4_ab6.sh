#!/bin/bash
# round 4: where the two-pass entropy stage loses its 6 ms: kernel traces of (a) split + plan, (b) split without the plan, (c) the one-pass kernel
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
export ZARC_GPU_LIB=$R/zarc_amd/libzarc_gpu_diag.so
for v in "A=1" "ZARC_GPU_ENT_NOPLAN=1" "ZARC_GPU_ENT_SPLIT=0"; do
  export $v; P=$R/gpurun_out/r4_ent_${v%%=*}; rm -rf $P; mkdir -p $P
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -- python3 $R/bench.py --entries 10000 --steps 2 --warmup 1 --no-cpu-baseline --no-host-path > $P/log 2>&1
  echo "== $v"; for f in $(find $P -name "*kernel_stats.csv"); do grep -E "zge_entropy|zge_plan|zge_match|blake3_chunks" $f | awk -F'","' '{print $1, $2, $4}' | cut -c1-60,200-; done
  unset ${v%%=*}
done
