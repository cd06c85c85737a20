#!/bin/bash
# Collects the rocprofv3 evidence for one round (run on the GPU box via gpurun):
#   kernel trace + stats of the bench command, then FETCH_SIZE and WRITE_SIZE in their own passes.
# Usage: tools/profile.sh <round-tag> [entries]
TAG=${1:-r01}; N=${2:-2048}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --entries $N --steps 2 --warmup 1 --no-cpu-baseline"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $CMD > $O/trace.log 2>&1; echo "trace rc=$?"
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $CMD > $O/fetch.log 2>&1; echo "fetch rc=$?"
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $CMD > $O/write.log 2>&1; echo "write rc=$?"
python3 - <<PY
import csv, glob, collections, json
out = {}
for name in ("fetch", "write"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % name, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            agg[k][0] += 1
            agg[k][1] += float(row["Counter_Value"])
    out[name] = {k: {"dispatches": v[0], "sum": v[1], "per_dispatch": v[1] / max(v[0], 1)} for k, v in agg.items()}
out["entries"] = $N
out["entry_bytes"] = 1 << 20
json.dump(out, open("$O/pmc_summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps({k: {kk: round(vv["per_dispatch"]) for kk, vv in v.items()} for k, v in out.items() if isinstance(v, dict)}, indent=0)[:1500])
PY
tail -1 $O/trace.log | cut -c1-600
