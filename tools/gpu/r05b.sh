set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05b; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_render_genre.py tests/test_gpu_callers.py tests/test_gpu_golden.py "tests/test_gpu_models.py" -x -q -m gpu -s > $OUT/A_pytest.log 2>&1; echo "pytest rc $?" >> $OUT/A_pytest.log
timeout 300 python tools/time_render_bm.py 32 > $OUT/B_time_bm.log 2>&1
timeout 900 python bench.py --no-train --cpu-seconds 3 --steps 10 > $OUT/C_bench.json 2> $OUT/C_bench.err
tail -15 $OUT/A_pytest.log; cat $OUT/B_time_bm.log; tail -c 1500 $OUT/C_bench.err; python - <<'PY'
import json,sys
try:
    l=[x for x in open(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/r05b/C_bench.json") if x.startswith("{")][-1]
    p=json.loads(l)
    print("value",p["value"]); print("hot",p["hot_path"]["shapes_per_s"],p["hot_path"]["ms_per_step"])
    print("roofline",json.dumps(p["roofline"])[:600])
    print("kernels",json.dumps(p["kernels"]))
    print("hot_path_batch1",json.dumps(p["hot_path_batch1"]))
    print("m1",json.dumps(p.get("m1"))[:900])
    print("b1",json.dumps(p["batch1"])[:700])
except Exception as e: print("no bench line",e)
PY
