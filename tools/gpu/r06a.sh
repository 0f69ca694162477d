#!/bin/bash
# round 6, first GPU call: the segment forward -- parity (new tests + the untouched renderer / camera / model parity files), timing A/B
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06a; mkdir -p $OUT
T0=$(date +%s)
timeout 1500 python -m pytest tests/test_gpu_render_seg.py tests/test_gpu_render.py tests/test_gpu_cam_bp.py tests/test_gpu_render_genre.py tests/test_gpu_golden.py tests/test_gpu_callers.py tests/test_gpu_models.py -x -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest rc $? in $(( $(date +%s) - T0 )) s" >> $OUT/pytest.log
for cfg in default 1,256 1,512 2,256 2,512; do
  if [ $cfg = default ]; then unset GENRE_SEG_CFG; else export GENRE_SEG_CFG=$cfg; fi
  timeout 600 python tools/time_render_seg.py > $OUT/time_$cfg.log 2>&1
done
unset GENRE_SEG_CFG
grep -E "passed|failed|FAILED|^E  |rc " $OUT/pytest.log | tail -30
for f in $OUT/time_*.log; do echo $f; grep "^batch" $f; tail -2 $f | grep -i "error\|Traceback"; done
