#!/bin/bash
# SQ / cache counters for the encoder kernels (separate passes; no tracing flags besides kernel names)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_match; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CMD="python3 $R/bench.py --entries 1024 --steps 1 --warmup 1 --no-cpu-baseline"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE"; do
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
for k in ("zarc_zge_match", "zarc_zge_entropy", "zarc_zstd_decode"):
    print(k)
    for c in sorted(agg[k]): print("   %-32s %.4g per dispatch (%d)" % (c, agg[k][c] / max(cnt[k][c], 1), cnt[k][c]))
PY
