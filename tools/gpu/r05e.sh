#!/bin/bash
# round 5, GPU call 5: occupancy words camera forward -> batch-minor renderer (tests, timing, bench), channels_last on the 2-D networks
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05e; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_cam_bp.py tests/test_gpu_render_genre.py tests/test_gpu_models.py tests/test_gpu_callers.py -x -q -m gpu > $OUT/A_pytest.log 2>&1; echo "rc $?" >> $OUT/A_pytest.log
timeout 600 python tools/time_render_bm.py 32 > $OUT/B_time_bm.log 2>&1
timeout 900 python bench.py --no-train --no-m1 --no-cpu-baseline --steps 10 > $OUT/D_bench.json 2> $OUT/D_bench.err
mkdir -p $ROOT/gpurun_out/miopen; cp -r $ROOT/genre-shapehd_amd/.miopen/* $ROOT/gpurun_out/miopen/
GENRE_MIOPEN_DIR=$ROOT/gpurun_out/miopen timeout 1200 python tools/m1_experiments.py > $OUT/C_m1x.log 2>&1
tail -12 $OUT/A_pytest.log; grep -v amdgpu.ids $OUT/B_time_bm.log; grep "M1X\|Error\|error" $OUT/C_m1x.log | cut -c1-300
python - <<'PY'
import json
try:
    l=[x for x in open("gpurun_out/r05e/D_bench.json") if x.startswith("{")][-1]
    p=json.loads(l); print("hot",p["hot_path"]["shapes_per_s"],p["hot_path"]["ms_per_step"]); print(json.dumps(p["kernels"])); print(json.dumps(p["roofline"])[:700])
except Exception as e: print("no bench line", e)
PY
tail -c 800 $OUT/D_bench.err
