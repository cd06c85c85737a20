#!/bin/bash
# kernel timeline (rocprofv3 --kernel-trace) of one configs[4]-shape pack + unpack: tools/trace_c5.sh lib tag
LIB=$1; TAG=${2:-c5}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ZARC_GPU_LIB=$R/$LIB timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --config c5 --gib 24 --steps 1 --warmup 1 --no-cpu-baseline --no-host-path > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
big = [r for r in rows if (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) > 2000000 and not r["Kernel_Name"].startswith("zarc_corpus")]
for r in big[-40:]:
    print("%-24s start %9.2f ms  dur %8.2f ms  stream-ish q=%s" % (r["Kernel_Name"].split("(")[0][:24], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Queue_Id", "?")))
PY
