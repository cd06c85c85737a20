#!/bin/bash
# register / scratch use of every kernel (CPU only: compiles each source to assembly and reads the code-object metadata).
# A latency-bound kernel that starts to spill gets slower without failing anything: run this after changing one.
cd "$(dirname "$0")/../zarc_amd/csrc" || exit 1
for f in *.hip; do
  fl=""; case $f in zge_entropy.hip|zstd_decode.hip) fl="-mllvm -amdgpu-sched-strategy=max-ilp";; zge_match.hip) fl="-mllvm -greedy-reverse-local-assignment=1 -mllvm -greedy-regclass-priority-trumps-globalness=1";; esac   # (the Makefile's FLAGS_*)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I. --cuda-device-only -S $fl -o /tmp/spills_$$.s $f 2>/dev/null || continue
  grep "\.name:\|\.private_segment_fixed_size\|\.vgpr_spill_count\|\.vgpr_count\|\.sgpr_spill_count\|\.group_segment_fixed_size" /tmp/spills_$$.s | paste - - - - - - |
    sed 's/  */ /g; s/\.group_segment_fixed_size/lds/; s/\.private_segment_fixed_size/scratch/; s/\.name: _Z[0-9]*\([a-z0-9_]*[a-z]\)[^ \t]*/\1/'
done
rm -f /tmp/spills_$$.s
