#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06j; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_render_seg.py tests/test_gpu_render.py -q -m gpu 2>&1 | grep -E "^E  .*(assert|Error)|passed|failed|FAILED" | cut -c1-250 | head -20
show() { grep "^batch" $1 | python -c "
import sys, json
for l in sys.stdin:
    b, _, j = l.partition('{'); r = json.loads('{' + j)
    print(b, {k: round(v, 1) for k, v in r.items() if k.startswith('seg')})"; }
timeout 600 python tools/time_render_seg.py > $OUT/time.log 2>&1; show $OUT/time.log
GENRE_HIP_LIB=$ROOT/tools/variants/libgenre_hip_tl.so timeout 300 python tools/seg_timeline.py 1 2>&1 | tee $OUT/timeline_b1.txt | grep -v "Warning\|amdgpu.ids" | head -12
