#!/bin/bash
# round 4: block slots per workgroup of the shared-table sequence kernel (64 / 32 / 16), diagnostic build
cd $GRAFT_REPO_ROOT
for w in 64 32 16; do for lf in 0 1; do echo -n "width=$w lit_first=$lf "; ZARC_GPU_SEQ_WIDTH=$w ZARC_GPU_LIT_FIRST=$lf bash tools/ab_unpack.sh zarc_amd/libzarc_gpu_diag.so; done; done
echo -n "product "; bash tools/ab_unpack.sh zarc_amd/libzarc_gpu.so
