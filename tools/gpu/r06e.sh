#!/bin/bash
# exact-waits staging: parity of the seg tests, timing, ablations, small-batch table variants
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06e; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_render_seg.py tests/test_gpu_render.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|^E  |rc " $OUT/pytest.log | tail -8
show() { grep "^batch" $1 | python -c "
import sys, json
for l in sys.stdin:
    b, _, j = l.partition('{'); r = json.loads('{' + j)
    print(b, {k: round(v, 1) for k, v in r.items() if k.startswith('seg')})"; }
for abl in 0 1 3; do
  GENRE_SEG_ABL=$abl timeout 600 python tools/time_render_seg.py > $OUT/time_abl$abl.log 2>&1; echo "abl $abl"; show $OUT/time_abl$abl.log
done
for cfg in 1,256 1,512; do
  GENRE_SEG_CFG=$cfg timeout 600 python tools/time_render_seg.py > $OUT/time_cfg$cfg.log 2>&1; echo "cfg $cfg"; show $OUT/time_cfg$cfg.log
done
export GENRE_TABLE_CACHE=0
for v in "8 256" "8 512" "16 512" "4 256"; do
  set -- $v
  GENRE_SEG_MAXSEG_SMALL=$1 GENRE_SEG_SPLIT_SMALL=$2 timeout 600 python tools/time_render_seg.py 2>&1 | grep "^batch 1 " > $OUT/time_small_$1_$2.log; echo "maxseg_small $1 split_small $2"; show $OUT/time_small_$1_$2.log
done
