#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06h; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_render_seg.py tests/test_gpu_render.py -x -q -m gpu 2>&1 | tail -3
show() { grep "^batch" $1 | python -c "
import sys, json
for l in sys.stdin:
    b, _, j = l.partition('{'); r = json.loads('{' + j)
    print(b, {k: round(v, 1) for k, v in r.items() if k.startswith('seg')})"; }
for cfg in default 1,256 1,512 1,1024 2,512 2,1024; do
  if [ $cfg = default ]; then unset GENRE_SEG_CFG; else export GENRE_SEG_CFG=$cfg; fi
  timeout 600 python tools/time_render_seg.py > $OUT/time_$cfg.log 2>&1; echo "cfg $cfg"; show $OUT/time_$cfg.log
done
unset GENRE_SEG_CFG
for B in 1 32; do GENRE_HIP_LIB=$ROOT/tools/variants/libgenre_hip_tl.so timeout 300 python tools/seg_timeline.py $B 2>&1 | tee $OUT/timeline_b$B.txt | grep -v "Warning\|amdgpu.ids"; done
GENRE_SEG_CFG=2,1024 GENRE_HIP_LIB=$ROOT/tools/variants/libgenre_hip_tl.so timeout 300 python tools/seg_timeline.py 32 2>&1 | tee $OUT/timeline_b32_21024.txt | grep -v "Warning\|amdgpu.ids"
