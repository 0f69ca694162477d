#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06g; mkdir -p $OUT
for B in 1 32; do GENRE_HIP_LIB=$ROOT/tools/variants/libgenre_hip_tl.so timeout 300 python tools/seg_timeline.py $B 2>&1 | tee $OUT/timeline_b$B.txt | grep -v Warning; done
