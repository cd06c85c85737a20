#!/bin/bash
# timeline of the LAST pack call of a bench run (kernels and copies, relative to its first kernel): tools/timeline_pack.sh <tag> [bench args]
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tlp_$TAG; mkdir -p $O
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
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "") , ""))
ev.sort()
m = [i for i, e in enumerate(ev) if e[2].startswith("zarc_zge_match")]
# the last pack call: from the last match kernel back to the previous unpack verdict / forward to the next zdec_count
last = m[-1]
lo = max([i for i, e in enumerate(ev[:last]) if e[2] in ("zarc_unpack_verdict", "zarc_corpus_fill")] + [0])
t0 = ev[last][0]
prev_end = None
for s, e, n, q in ev[lo:]:
    if n == "zarc_zdec_count" and s > t0: break
    gap = "" if prev_end is None or s - prev_end < 200000 else "   <-- %.3f ms idle" % ((s - prev_end) / 1e6)
    if (e - s) > 20000 or gap: print("%9.3f ms  +%8.3f ms  %-28s %s%s" % ((s - t0) / 1e6, (e - s) / 1e6, n, q, gap))
    prev_end = max(prev_end or 0, e)
PY
