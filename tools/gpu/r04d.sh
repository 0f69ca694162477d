#!/bin/bash
# round 4, GPU call 4: tests again (train statements, skip test, thin convs incl. the critic), the sampler that saves nothing for
# blocked tiles (A/B worker + bm tests), bench with all train configs, WGAN-GP kernel trace
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04d; mkdir -p "$OUT"
export TMPDIR=/tmp
MI=$ROOT/genre-shapehd_amd/.miopen
timeout 1800 python -m pytest tests/test_gpu_render.py tests/test_gpu_render_genre.py tests/test_gpu_thin_conv.py tests/test_gpu_z_train.py -q -m gpu --tb=short -s 2>&1 | cut -c1-500 > "$OUT/A_pytest.log"
timeout 400 python tools/ab_round4.py --worker bm default > "$OUT/B_bm.log" 2>&1
timeout 1500 python bench.py --steps 10 --train-steps 4 --train-configs all --cpu-seconds 5 > "$OUT/C_bench.json" 2> "$OUT/C_bench.err"
cd /tmp
export MIOPEN_USER_DB_PATH=$MI/db MIOPEN_CUSTOM_CACHE_DIR=$MI/cache
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof_w" -o t -- python "$ROOT/genre-shapehd_amd/train.py" --config wgangp --batch 8 --steps 6 > "$OUT/F_train_wgangp.log" 2>&1
DB=$(ls "$OUT"/prof_w/t_results.db "$OUT"/prof_w/*/t_results.db 2>/dev/null | head -1)
python "$ROOT/profiles/summarize_rocpd.py" "$DB" > "$OUT/F_train_wgangp_kernel_stats.txt" 2>&1; rm -rf "$OUT/prof_w"
cd "$ROOT"
tar czf "$OUT/miopen_after.tgz" -C "$ROOT/genre-shapehd_amd" .miopen
grep -E "passed|failed|FAILED|^E  " "$OUT/A_pytest.log" | head -40; grep AB4 "$OUT/B_bm.log"; tail -c 1500 "$OUT/C_bench.json"; head -12 "$OUT/F_train_wgangp_kernel_stats.txt" | cut -c1-170
