#!/bin/bash
# ablations of seg_sample_kernel at batch 1 / 32 (GENRE_SEG_ABL bits: 1 no march, 2 no tile loads, 4 return at once)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06d; mkdir -p $OUT
for abl in 0 1 2 3 4; do
  GENRE_SEG_ABL=$abl timeout 600 python tools/time_render_seg.py > $OUT/time_abl$abl.log 2>&1
  echo "abl $abl"; grep "^batch" $OUT/time_abl$abl.log | python -c "
import sys, json
for l in sys.stdin:
    b, _, j = l.partition('{'); r = json.loads('{' + j)
    print(b, {k: round(v, 1) for k, v in r.items() if k.startswith('seg')})"
done
