#!/bin/bash
# Collects the rocprofv3 evidence for one round (run on the GPU box via gpurun):
#   kernel trace + stats of the bench command, then FETCH_SIZE and WRITE_SIZE in their own passes.
# Usage: tools/profile.sh <round-tag> [entries]      (BENCH_ARGS="--config c4 --gib 8": another workload instead of --entries N)
TAG=${1:-r01}; N=${2:-2048}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py ${BENCH_ARGS:---entries $N} --steps 2 --warmup 1 --no-cpu-baseline --no-host-path"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $CMD > $O/trace.log 2>&1; echo "trace rc=$?"
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $CMD > $O/fetch.log 2>&1; echo "fetch rc=$?"
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $CMD > $O/write.log 2>&1; echo "write rc=$?"
# SQ counters of the dominant kernels (issue / wait split; each group in its own pass)
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1)); timeout 900 rocprofv3 --pmc $set --output-format csv -d $O/sq$i -- $CMD > $O/sq$i.log 2>&1; echo "sq$i rc=$?"
done
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
out["bench_args"] = "${BENCH_ARGS:-}"   # non-empty: another workload than entries x 1 MiB (bench.py quotes roofline.traffic for the default workload only)
import hashlib
out["lib_sha16"] = hashlib.sha256(open("$R/zarc_amd/libzarc_gpu.so", "rb").read()).hexdigest()[:16]
import sys
sys.path.insert(0, "$R")
import bench
out["src_sha16"] = bench.src_sha16()
sq = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("$O/sq*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if k.startswith("zarc_") and k != "zarc_corpus_fill":
            sq[k][row["Counter_Name"]][0] += 1; sq[k][row["Counter_Name"]][1] += float(row["Counter_Value"])
out["sq_per_dispatch"] = {k: {c: v[1] / max(v[0], 1) for c, v in d.items()} for k, d in sq.items()}
json.dump(out, open("$O/pmc_summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps({k: {kk: round(vv["per_dispatch"]) for kk, vv in v.items()} for k, v in out.items() if k in ("fetch", "write")}, indent=0)[:1500])
PY
tail -1 $O/trace.log | cut -c1-600
