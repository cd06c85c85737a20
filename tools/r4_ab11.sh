#!/bin/bash
# round 4, timing: the match finder's input loads non-temporal (v5), its sequence stores as well (v6)
cd $GRAFT_REPO_ROOT
N=10000 bash tools/ab.sh zarc_amd/libzarc_gpu.so zarc_amd/csrc/ab/v5.so zarc_amd/csrc/ab/v6.so zarc_amd/libzarc_gpu.so
