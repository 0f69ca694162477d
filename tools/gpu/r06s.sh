#!/bin/bash
# quick: segment backward numerics on sharp volumes + the fused-renderer bench rows
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06s; mkdir -p $OUT
python tools/debug_seg_bwd.py sharp pad 2>&1 | grep -v amdgpu.ids | grep -E "grad diff|new vs exact"
timeout 1500 python bench.py --no-train --no-m1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python - <<'PY'
import json
l = [x for x in open("gpurun_out/r06s/bench.json") if x.startswith("{")]
j = json.loads(l[-1])
for k, v in j["kernels"].items():
    if "fused" in k and ("bwd" in k or "grad" in k): print("  ", k, v["us"])
print("hot_path_batch1", {k: (v.get("us_per_image_fwd_bwd") if isinstance(v, dict) else v) for k, v in j["hot_path_batch1"].items() if k != "what"})
PY
