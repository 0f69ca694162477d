#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06o; mkdir -p $OUT
GENRE_HIP_LIB=$ROOT/tools/variants/libgenre_hip_tlb.so timeout 600 python tools/bm_timeline.py 32 2>&1 | tee $OUT/bm_timeline_b32.txt | grep -v "Warning\|amdgpu.ids"
