#!/bin/bash
# round 4, GPU call 5: the halo-form backward (tests + timing against the pull form), sampler tweaks, skip test
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04e; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python tools/ab_round4.py --worker bm default > "$OUT/B_bm.log" 2>&1
timeout 1500 python -m pytest tests/test_gpu_render.py tests/test_gpu_render_genre.py -q -m gpu --tb=short -s 2>&1 | cut -c1-400 > "$OUT/A_pytest.log"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o t -- python "$ROOT/tools/ab_round4.py" --worker bm prof > "$OUT/C_prof.log" 2>&1
DB=$(ls "$OUT"/prof/t_results.db "$OUT"/prof/*/t_results.db 2>/dev/null | head -1)
python "$ROOT/profiles/summarize_rocpd.py" "$DB" > "$OUT/C_kernel_stats.txt" 2>&1; rm -rf "$OUT/prof"
cd "$ROOT"
grep AB4 "$OUT/B_bm.log"; grep -E "passed|failed|FAILED|^E  |halo vs" "$OUT/A_pytest.log" | head -30; head -14 "$OUT/C_kernel_stats.txt" | cut -c1-150
