#!/bin/bash
# per-kernel average times of one bench run under rocprofv3 --kernel-trace --stats: tools/ktrace.sh <tag> [lib]   (env passes through)
TAG=$1; LIB=${2:-zarc_amd/libzarc_gpu.so}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kt_$TAG; mkdir -p $O
export ZARC_GPU_LIB=$R/$LIB
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --entries ${N:-10000} --steps 2 --warmup 1 --no-cpu-baseline > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$O/**/*_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].split("(")[0]
        if n.startswith("zarc_") and n != "zarc_corpus_fill": print("%-22s %3s x %9.3f ms" % (n, r["Calls"], float(r["AverageNs"]) / 1e6))
PY
