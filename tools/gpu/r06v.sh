#!/bin/bash
# per-kernel durations of the batch-1 chain (forward + backward), GenRe's volume and the live-gradient variant
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06v; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for S in 50 0.9; do
rocprofv3 --kernel-trace --stats -d "$OUT/prof_$S" -o t -- python "$ROOT/tools/prof_b1_chain.py" $S > /dev/null 2> "$OUT/prof.err"
python "$ROOT/profiles/summarize_rocpd.py" $(ls "$OUT"/prof_$S/t_results.db "$OUT"/prof_$S/*/t_results.db 2>/dev/null | head -1) > "$OUT/kernel_stats_b1_s$S.txt" 2>&1
rm -rf "$OUT/prof_$S"
echo "== pre_scale $S"; head -24 "$OUT/kernel_stats_b1_s$S.txt" | cut -c1-90,100-150
done
