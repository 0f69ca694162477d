#!/bin/bash
# full GPU suite under -x + smoke + the bench line (no train, to keep it short) on the segment-forward tree
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06k; mkdir -p $OUT
T0=$(date +%s)
timeout 2400 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $? in $(( $(date +%s) - T0 )) s" >> $OUT/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?" >> $OUT/smoke.log
timeout 1500 python bench.py --no-train > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?" >> $OUT/bench.err
grep -E "passed|failed|FAILED|^E  |rc " $OUT/pytest_gpu.log | tail -12; tail -3 $OUT/smoke.log; tail -3 $OUT/bench.err
python - <<'PY'
import json
l = [x for x in open("gpurun_out/r06k/bench.json") if x.startswith("{")]
if l:
    j = json.loads(l[-1])
    print("value", j["value"], "hot_path", j["hot_path"]["shapes_per_s"], j["hot_path"]["ms_per_step"])
    print("roofline", {k: j["roofline"][k] for k in ("bound", "kernel", "frac", "avg_launch_us", "traffic") if k in j["roofline"]})
    for k, v in j["kernels"].items():
        print("  ", k, v)
    print("batch1", {k: v for k, v in j["batch1"].items() if "us" in k})
    print("hot_path_batch1", j["hot_path_batch1"])
    print("forward_only", j["forward_only"])
    print("m2", j["roofline_m2"])
PY
