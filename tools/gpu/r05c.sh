#!/bin/bash
# round 5, GPU call 3: (A) forward bricks 4x4x8 (512-thread workgroups, four per CU) against 4x8x8; (B) the GenRe forward with the
# inference rewrites (BatchNorm folding, sub-pixel transposed convolutions)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05c; mkdir -p $OUT
LIB=$ROOT/genre-shapehd_amd/csrc/libgenre_hip.so
cp $LIB /tmp/base.so
for v in base by4; do
  if [ $v = by4 ]; then cp $ROOT/tools/variants/libgenre_hip_by4.so $LIB; export GENRE_BM_BY=4; else export GENRE_BM_BY=8; fi
  timeout 600 python tools/time_render_bm.py 32 > $OUT/A_time_$v.log 2>&1
  timeout 600 python -m pytest tests/test_gpu_render.py tests/test_gpu_render_genre.py -x -q -m gpu -k "batch_minor" > $OUT/A_pytest_$v.log 2>&1; echo "rc $?" >> $OUT/A_pytest_$v.log
done
cp /tmp/base.so $LIB; export GENRE_BM_BY=8
mkdir -p $ROOT/gpurun_out/miopen; cp -r $ROOT/genre-shapehd_amd/.miopen/* $ROOT/gpurun_out/miopen/
GENRE_MIOPEN_DIR=$ROOT/gpurun_out/miopen timeout 1500 python tools/m1_experiments.py > $OUT/B_m1x.log 2>&1
for v in base by4; do echo "== $v"; grep -v amdgpu.ids $OUT/A_time_$v.log; tail -3 $OUT/A_pytest_$v.log; done
grep "M1X\|Error\|error" $OUT/B_m1x.log | cut -c1-400
