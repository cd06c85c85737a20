#!/bin/bash
# timeline of the LAST unpack call of a bench run: kernels and copies with start / duration relative to the first decoder kernel
# tools/timeline.sh <tag> [bench args]
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tl_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-path "$@" > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
ev = []
for f in glob.glob("$O/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], "q%s" % r.get("Queue_Id", "")))
for f in glob.glob("$O/**/*_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", ""), ""))
ev.sort()
last = max(i for i, e in enumerate(ev) if e[2] == "zarc_zdec_count")
t0 = ev[last][0]
for s, e, n, q in ev[last:]:
    if n == "zarc_corpus_fill": break
    print("%9.3f ms  +%8.3f ms  %-28s %s" % ((s - t0) / 1e6, (e - s) / 1e6, n, q))
PY
