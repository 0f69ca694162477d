#!/bin/bash
# segment forward, second pass: parity + timing + kernel trace
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06c; mkdir -p $OUT
T0=$(date +%s)
timeout 1500 python -m pytest tests/test_gpu_render_seg.py tests/test_gpu_render.py tests/test_gpu_cam_bp.py tests/test_gpu_render_genre.py tests/test_gpu_golden.py tests/test_gpu_callers.py tests/test_gpu_models.py -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest rc $? in $(( $(date +%s) - T0 )) s" >> $OUT/pytest.log
for cfg in default 1,256 2,256; do
  if [ $cfg = default ]; then unset GENRE_SEG_CFG; else export GENRE_SEG_CFG=$cfg; fi
  timeout 600 python tools/time_render_seg.py > $OUT/time_$cfg.log 2>&1
done
unset GENRE_SEG_CFG
cd /tmp && export TMPDIR=/tmp
for B in 1 32; do
  rocprofv3 --kernel-trace --stats -d $OUT/trace_b$B -o t -- python $ROOT/tools/prof_seg.py $B > $OUT/prof_b$B.log 2>&1
  python $ROOT/profiles/summarize_rocpd.py $OUT/trace_b$B/t_results.db > $OUT/kernel_stats_b$B.txt 2>&1
  rm -rf $OUT/trace_b$B
  echo "== batch $B"; head -7 $OUT/kernel_stats_b$B.txt | cut -c1-60,100-200
done
cd $ROOT
grep -E "passed|failed|FAILED|^E  |rc " $OUT/pytest.log | tail -30
for f in $OUT/time_*.log; do echo $f; grep "^batch" $f; tail -2 $f | grep -i "error\|Traceback"; done
