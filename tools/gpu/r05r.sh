#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05r; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_render_genre.py -x -q -m gpu > $OUT/A_pytest.log 2>&1; echo "rc $?" >> $OUT/A_pytest.log; tail -3 $OUT/A_pytest.log
timeout 900 python bench.py --no-train --no-m1 --no-cpu-baseline --steps 10 > $OUT/D_bench.json 2> $OUT/D_bench.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r05r/D_bench.json") if x.startswith("{")][-1]
p=json.loads(l); print("hot",p["hot_path"]["shapes_per_s"],p["hot_path"]["ms_per_step"]); print(json.dumps({k:v for k,v in p["kernels"].items() if "bwd_bm" in k}))
PY
