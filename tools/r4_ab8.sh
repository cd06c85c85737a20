#!/bin/bash
# round 4: far inserts as stores instead of atomics (timing only: the order of same-bucket inserts of one tile is then undefined), levels 3 and 9
cd $GRAFT_REPO_ROOT
N=10000 bash tools/ab.sh zarc_amd/libzarc_gpu.so zarc_amd/csrc/ab/v1.so zarc_amd/csrc/ab/v2.so zarc_amd/libzarc_gpu.so
GIB=8 bash tools/ab_c4.sh zarc_amd/csrc/ab/v1.so zarc_amd/csrc/ab/v2.so
