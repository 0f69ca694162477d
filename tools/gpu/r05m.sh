#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05m; mkdir -p $OUT
mkdir -p $ROOT/gpurun_out/miopen; cp -r $ROOT/genre-shapehd_amd/.miopen/* $ROOT/gpurun_out/miopen/
export GENRE_MIOPEN_DIR=$ROOT/gpurun_out/miopen
timeout 300 python tools/tune_heavy_convs.py > $OUT/1_before.log 2>&1
MIOPEN_FIND_ENFORCE=3 timeout 900 python tools/tune_heavy_convs.py > $OUT/2_tune.log 2>&1; echo "tune rc $?" >> $OUT/2_tune.log
timeout 300 python tools/tune_heavy_convs.py > $OUT/3_after.log 2>&1
grep -h "TUNE\|rc" $OUT/1_before.log $OUT/2_tune.log $OUT/3_after.log
