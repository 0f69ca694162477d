#!/bin/bash
# segment-form backward of the standard layout: parity (renderer gradient tests, models, train) + bench
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06p; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_render_seg.py tests/test_gpu_render.py tests/test_gpu_render_genre.py tests/test_gpu_models.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | grep -E "^E  .*(assert|Error)|passed|failed|FAILED" | cut -c1-300 | head
timeout 1500 python bench.py --no-train --no-m1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; tail -3 $OUT/bench.err
python - <<'PY'
import json
l = [x for x in open("gpurun_out/r06p/bench.json") if x.startswith("{")]
j = json.loads(l[-1])
print("hot_path", j["hot_path"]["shapes_per_s"], j["hot_path"]["ms_per_step"])
for k, v in j["kernels"].items():
    if "fused" in k: print("  ", k, v)
print("hot_path_batch1", {k: (v.get("us_per_image_fwd_bwd") if isinstance(v, dict) else v) for k, v in j["hot_path_batch1"].items() if k != "what"})
print("batch1", {k: v for k, v in j["batch1"].items() if "us" in k})
PY
