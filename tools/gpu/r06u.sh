#!/bin/bash
# kernel trace of the segment backward alone (soft volume), batch 32 and 1
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06u; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for B in 32 1; do
rocprofv3 --kernel-trace --stats -d "$OUT/prof_$B" -o t -- python "$ROOT/tools/time_seg_bwd.py" $B > /dev/null 2> "$OUT/prof.err"
python "$ROOT/profiles/summarize_rocpd.py" $(ls "$OUT"/prof_$B/t_results.db "$OUT"/prof_$B/*/t_results.db 2>/dev/null | head -1) > "$OUT/kernel_stats_b$B.txt" 2>&1
rm -rf "$OUT/prof_$B"
echo "== batch $B"; grep -E "seg_|name" "$OUT/kernel_stats_b$B.txt" | cut -c1-70,100-160
done
