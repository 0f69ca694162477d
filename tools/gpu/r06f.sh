#!/bin/bash
# finer ablations of the sampler's skeleton (GENRE_SEG_ABL bits: 1 no march, 2 no tile loads, 4 return at once, 8 no (P,S) stores,
# 16 no dependent dirs gather, 32 no LDS tile stores, 64 return after one dependent load, 128 return at the barrier)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06f; mkdir -p $OUT
show() { grep "^batch" $1 | python -c "
import sys, json
for l in sys.stdin:
    b, _, j = l.partition('{'); r = json.loads('{' + j)
    print(b, {k: round(v, 1) for k, v in r.items() if k.startswith('seg')})"; }
for abl in 0 4 64 128 129 137 11 27 59 3; do
  GENRE_SEG_ABL=$abl timeout 600 python tools/time_render_seg.py > $OUT/time_abl$abl.log 2>&1; echo "abl $abl"; show $OUT/time_abl$abl.log
done
timeout 300 python -m pytest tests/test_gpu_render_seg.py -x -q -m gpu 2>&1 | tail -3
