#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06i; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_render_seg.py tests/test_gpu_render.py -x -q -m gpu 2>&1 | tail -3
show() { grep "^batch" $1 | python -c "
import sys, json
for l in sys.stdin:
    b, _, j = l.partition('{'); r = json.loads('{' + j)
    print(b, {k: round(v, 1) for k, v in r.items() if k.startswith('seg')})"; }
for cfg in default 1,256; do
  if [ $cfg = default ]; then unset GENRE_SEG_CFG; else export GENRE_SEG_CFG=$cfg; fi
  timeout 600 python tools/time_render_seg.py > $OUT/time_$cfg.log 2>&1; echo "cfg $cfg"; show $OUT/time_$cfg.log
done
unset GENRE_SEG_CFG
export GENRE_TABLE_CACHE=0
for v in "8 256" "8 512" "12 256"; do
  set -- $v
  GENRE_SEG_MAXSEG_SMALL=$1 GENRE_SEG_SPLIT_SMALL=$2 timeout 600 python tools/time_render_seg.py 2>&1 | grep "^batch 1 " > $OUT/time_small_$1_$2.log; echo "maxseg_small $1 split_small $2"; show $OUT/time_small_$1_$2.log
done
unset GENRE_TABLE_CACHE
GENRE_HIP_LIB=$ROOT/tools/variants/libgenre_hip_tl.so timeout 300 python tools/seg_timeline.py 1 2>&1 | tee $OUT/timeline_b1.txt | grep -v "Warning\|amdgpu.ids"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace_b1 -o t -- python $ROOT/tools/prof_seg.py 1 > $OUT/prof_b1.log 2>&1
python $ROOT/profiles/summarize_rocpd.py $OUT/trace_b1/t_results.db > $OUT/kernel_stats_b1.txt 2>&1; rm -rf $OUT/trace_b1
head -7 $OUT/kernel_stats_b1.txt | cut -c1-60,100-200
