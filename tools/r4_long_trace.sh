#!/bin/bash
# kernel times (and SQ counters) of unpacking libzstd's frames: the frames are made first, without the profiler
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kt_long; mkdir -p $O
cd $R && python3 tools/libzstd_frames_cache.py make /tmp/lz3.pkl 2048 3 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/libzstd_frames_cache.py run /tmp/lz3.pkl 3 > $O/log.txt 2>&1
tail -1 $O/log.txt
python3 - <<PY
import csv, glob
for f in glob.glob("$O/trace/**/*_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].split("(")[0]
        if n.startswith("zarc_z"): print("%-26s %3s x %9.3f ms  (min %.3f max %.3f)" % (n, r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
PY
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU"; do
  i=$((i+1)); timeout 600 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/tools/libzstd_frames_cache.py run /tmp/lz3.pkl 1 > $O/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for k in ("zarc_zdec_seqs_lds", "zarc_zdec_seqs_shared"):
    print(k)
    for c in sorted(agg[k]): print("   %-28s %.4g per dispatch" % (c, agg[k][c] / max(cnt[k][c], 1)))
PY
