#!/bin/bash
# round 4, GPU call 2: new tests (train statements, thin convs, zero-mask skip), batch-1 / scatter A/B, FETCH_SIZE probe,
# train steps with the thin convolutions (kernel traces)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04b; mkdir -p "$OUT"
export TMPDIR=/tmp
MI=$ROOT/genre-shapehd_amd/.miopen
timeout 1500 python -m pytest tests/test_gpu_render.py tests/test_gpu_render_genre.py tests/test_gpu_thin_conv.py tests/test_gpu_z_train.py -q -m gpu -s 2>&1 | grep -v "^$" | tail -120 > "$OUT/A_pytest.log"
timeout 1500 python tools/ab_round4.py --cam-tests > "$OUT/B_ab.log" 2>&1
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/probe" -o fetch -- "$ROOT/tools/fetch_size_bench" > "$OUT/C_probe.log" 2>&1
F=$(ls "$OUT"/probe/fetch_results.db "$OUT"/probe/*/fetch_results.db 2>/dev/null | head -1)
python "$ROOT/profiles/fetch_probe_table.py" "$F" > "$OUT/C_fetch_probe.txt" 2>&1; rm -rf "$OUT/probe"
export MIOPEN_USER_DB_PATH=$MI/db MIOPEN_CUSTOM_CACHE_DIR=$MI/cache
for cfg in "shapehd 8" "genre 4"; do
  set -- $cfg
  timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$1" -o t -- python "$ROOT/genre-shapehd_amd/train.py" --config $1 --batch $2 --steps 8 > "$OUT/F_train_$1.log" 2>&1
  DB=$(ls "$OUT"/prof_$1/t_results.db "$OUT"/prof_$1/*/t_results.db 2>/dev/null | head -1)
  python "$ROOT/profiles/summarize_rocpd.py" "$DB" > "$OUT/F_train_${1}_kernel_stats.txt" 2>&1
  rm -rf "$OUT/prof_$1"
done
cd "$ROOT"
tar czf "$OUT/miopen_after.tgz" -C "$ROOT/genre-shapehd_amd" .miopen
grep -E "passed|failed|FAILED" "$OUT/A_pytest.log"; grep AB4 "$OUT/B_ab.log"; cat "$OUT/C_fetch_probe.txt"; head -8 "$OUT"/F_train_shapehd_kernel_stats.txt
