#!/bin/bash
# kernel trace of the segment forward at batch 1 / 8 / 32
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06b; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for B in 1 32; do
  rocprofv3 --kernel-trace --stats -d $OUT/trace_b$B -o t -- python $ROOT/tools/prof_seg.py $B > $OUT/prof_b$B.log 2>&1
  f=$(find $OUT/trace_b$B -name "*kernel_stats.csv" | head -1)
  echo "== batch $B"; head -12 "$f" | cut -c1-200
done
