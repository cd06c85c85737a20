#!/bin/bash
# timing-only experiments on the match finder: ablation bits and the match-length cap
run() { timeout -k 10 300 python bench.py --entries ${N:-10000} --steps 1 --warmup 1 --no-cpu-baseline --kind ${KIND:--1} 2>/dev/null | python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]) if l else None; print(d['kernel_ms']['zge_match'], d['ratio']) if d else print('fail')"; }
for d in 0 1 2 4 8 16 32 64; do echo -n "dbg=$d "; ZARC_GPU_DBG=$d run; done
for c in 32 64 128; do echo -n "cap=$c "; ZARC_GPU_CAP=$c run; done
