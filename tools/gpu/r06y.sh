#!/bin/bash
# image-minor renderer rows of the bench + its parity tests (A/B of sampler changes: GENRE_HIP_LIB=... for the other library)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06y; mkdir -p $OUT
timeout 1500 python bench.py --no-train --no-m1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python - <<'PY'
import json
l = [x for x in open("gpurun_out/r06y/bench.json") if x.startswith("{")]
j = json.loads(l[-1])
print("hot_path", round(j["hot_path"]["shapes_per_s"]), j["hot_path"]["ms_per_step"])
for k, v in j["kernels"].items():
    if "_bm" in k: print("  ", k, v["us"], v.get("us_in_step_order", ""))
PY
