#!/bin/bash
# per-kernel durations of the renderer phases (genre / dense / soft) from one kernel trace of profiles/pmc_targets.py
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06r; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/prof_t" -o targets -- python "$ROOT/profiles/pmc_targets.py" 32 16 > /dev/null 2> "$OUT/prof_t.err"
python "$ROOT/profiles/summarize_rocpd.py" --phases $(ls "$OUT"/prof_t/targets_results.db "$OUT"/prof_t/*/targets_results.db 2>/dev/null | head -1) > "$OUT/kernel_stats_phases.txt" 2>&1
rm -rf "$OUT/prof_t"
grep -E "seg_|render_bwd_brick|name|publish" "$OUT/kernel_stats_phases.txt" | cut -c1-220
