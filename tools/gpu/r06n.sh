#!/bin/bash
# batch-minor sampler: rows carry brick coordinates, everything the row determines requested at once (HINT template): parity + bench
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06n; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_render.py tests/test_gpu_render_genre.py tests/test_gpu_render_seg.py tests/test_gpu_models.py -x -q -m gpu 2>&1 | tail -3
timeout 1500 python bench.py --no-train --no-m1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python - <<'PY'
import json
l = [x for x in open("gpurun_out/r06n/bench.json") if x.startswith("{")]
j = json.loads(l[-1])
print("hot_path", j["hot_path"]["shapes_per_s"], j["hot_path"]["ms_per_step"])
for k, v in j["kernels"].items():
    if "bm" in k or "fused" in k: print("  ", k, v)
print("hot_path_batch1", {k: (v.get("us_per_image_fwd_bwd") if isinstance(v, dict) else v) for k, v in j["hot_path_batch1"].items() if k != "what"})
print("roofline", {k: j["roofline"].get(k) for k in ("bound", "kernel", "frac", "frac_on_bytes_needed", "avg_launch_us", "traffic")})
PY
