#!/bin/bash
# round 5, GPU call 6: tile words + per-segment constants (tests), then the evidence run (profiles/collect_pmc.sh r05f)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05f; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_cam_bp.py tests/test_gpu_render_genre.py tests/test_gpu_callers.py -x -q -m gpu > $OUT/A_pytest.log 2>&1; rc=$?; echo "rc $rc" >> $OUT/A_pytest.log
tail -5 $OUT/A_pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout 300 python tools/time_render_bm.py 32 2>&1 | grep -v amdgpu
bash profiles/collect_pmc.sh r05f 32 > $OUT/collect.log 2>&1
tail -60 $OUT/collect.log
