#!/bin/bash
# round 4, decoder A/B on the GPU box: shared-table sequence kernel (default), literals launched first, shared kernel off; then the level tiers
cd $GRAFT_REPO_ROOT; O=gpurun_out
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-host-path --steps 3 --warmup 1 2>$O/r4_ab1_$tag.err | tail -1 > $O/r4_ab1_$tag.json; python - $O/r4_ab1_$tag.json $tag <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], "pack", d["value"], "unpack", d["unpack_gibs"], d["roundtrip_bit_exact"], "ratio", d["ratio"], d["kernel_ms"], d["unpack_kernel_ms"])
PY
}
run default A=1
run litfirst ZARC_GPU_LIB=$PWD/zarc_amd/libzarc_gpu_diag.so ZARC_GPU_LIT_FIRST=1
run noshared ZARC_GPU_LIB=$PWD/zarc_amd/libzarc_gpu_diag.so ZARC_GPU_SEQ_SHARED=0
timeout -k 10 300 python bench.py --level -1 --no-host-path --steps 3 --warmup 1 2>$O/r4_lm1.err | tail -1 > $O/r4_bench_level_m1.json
python -c "
import json; d=json.load(open('$O/r4_bench_level_m1.json')); print('level -1: pack', d['value'], 'unpack', d['unpack_gibs'], 'ratio', d['ratio'], 'vs ref', d.get('ratio_vs_reference'), d.get('ratio_reference'), d['kernel_ms'])"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "level_tiers or pack_bit_exact" 2>&1 | tail -3
