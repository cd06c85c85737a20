#!/bin/bash
# SQ counters for the decoder kernels (separate passes; program directly after --)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_decode; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export ZARC_GPU_DEC_SIDE=0
CMD="python3 $R/bench.py --entries ${N:-4096} --steps 1 --warmup 1 --no-cpu-baseline"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_IFETCH" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- $CMD > $O/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for k in ("zarc_zdec_seqs_shared", "zarc_zdec_seqs", "zarc_zstd_frames", "zarc_zdec_literals", "zarc_zstd_decode"):
    print(k)
    for c in sorted(agg[k]): print("   %-32s %.4g per dispatch (%d)" % (c, agg[k][c] / max(cnt[k][c], 1), cnt[k][c]))
PY
