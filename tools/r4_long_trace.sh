#!/bin/bash
# kernel times of unpacking libzstd's frames (tools/libzstd_frames_rate.py) under rocprofv3 --kernel-trace --stats
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kt_long; mkdir -p $O
cd /tmp && export TMPDIR=/tmp ZARC_TOOL_THREADS=1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/libzstd_frames_rate.py 1024 3 > $O/log.txt 2>&1
tail -3 $O/log.txt
python3 - <<PY
import csv, glob
for f in glob.glob("$O/**/*_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].split("(")[0]
        if n.startswith("zarc_z"): print("%-26s %3s x %9.3f ms  (min %.3f max %.3f)" % (n, r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
PY
