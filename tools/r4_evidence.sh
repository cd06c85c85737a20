#!/bin/bash
# round 4 evidence batch on the GPU box: tests, the driver-shaped bench line, kernel trace + PMC passes, the other configurations
TAG=${1:-r04}; cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$? $(tail -1 $O/gpu_tests.log)"
timeout -k 10 600 python bench.py --steps 5 --warmup 2 > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$?"
bash tools/profile.sh $TAG 10000 > $O/profile.log 2>&1; echo "profile rc=$?"; tail -3 $O/profile.log | cut -c1-400
timeout -k 10 600 python bench.py --config c5 --gib 24 --steps 3 --warmup 1 --no-host-path > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?"
timeout -k 10 600 python bench.py --config small --steps 3 --warmup 1 --no-host-path > $O/bench_small.json 2> $O/bench_small.err; echo "small rc=$?"
timeout -k 10 600 python bench.py --level -1 --steps 3 --warmup 1 --no-host-path > $O/bench_level_m1.json 2> $O/bench_level_m1.err; echo "level -1 rc=$?"
timeout -k 10 900 python bench.py --config c4 --gib 64 --steps 1 --warmup 1 --no-host-path > $O/bench_c4_64g.json 2> $O/bench_c4_64g.err; echo "c4 64g rc=$?"
BENCH_ARGS="--config c4 --gib 8" bash tools/profile.sh ${TAG}c4 > $O/profile_c4.log 2>&1; echo "profile c4 rc=$?"
for S in 4096 16384 65536; do bash tools/ab_size.sh $((2147483648 / S)) $S zarc_amd/libzarc_gpu.so; done > $O/entry_sizes.log 2>&1; cat $O/entry_sizes.log
for f in bench_full bench_c5 bench_small bench_level_m1 bench_c4_64g; do python - $O/$f.json $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], "pack", d["value"], "unpack", d["unpack_gibs"], "ratio", d["ratio"], d.get("ratio_vs_reference"), d["roundtrip_bit_exact"], d["kernel_ms"], d["unpack_kernel_ms"])
except Exception as e: print(sys.argv[2], "unreadable", e)
PY
done
