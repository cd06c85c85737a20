#!/bin/bash
# round 4: how many waves does the shared-table kernel turn down at configs[1]?  then SQ counters of the decoder kernels
cd $GRAFT_REPO_ROOT; O=gpurun_out
ZARC_GPU_LIB=$PWD/zarc_amd/libzarc_gpu_diag.so ZARC_GPU_DEC_STATS=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-host-path --steps 1 --warmup 1 2>&1 >/dev/null | grep -E "zdec_seqs|zstd_decode" | sort | uniq -c
R=$GRAFT_REPO_ROOT; P=$R/gpurun_out/r4_pmc_dec; mkdir -p $P; cd /tmp; export TMPDIR=/tmp
CMD="python3 $R/bench.py --entries 10000 --steps 1 --warmup 1 --no-cpu-baseline --no-host-path"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $P/p$i -- $CMD > $P/p$i.log 2>&1 || echo "pass $i failed"
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- $CMD > $P/trace.log 2>&1 || echo "trace failed"
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$P/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for k in ("zarc_zdec_seqs_shared", "zarc_zdec_seqs", "zarc_zdec_literals", "zarc_zge_entropy_p1", "zarc_zge_entropy_p2", "zarc_zge_plan"):
    print(k)
    for c in sorted(agg[k]): print("   %-32s %.4g per dispatch (%d)" % (c, agg[k][c] / max(cnt[k][c], 1), cnt[k][c]))
for f in glob.glob("$P/trace/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:3000])
PY
