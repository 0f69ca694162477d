#!/bin/bash
# round 4, GPU call 1: reproduce the f-3 gradient failure under the shipped MIOpen find-db, name the solver, profile the train steps
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04a; mkdir -p "$OUT"
export TMPDIR=/tmp
MI=$ROOT/genre-shapehd_amd/.miopen
# A: the three train tests under the shipped db (no -x)
timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu 2>&1 | grep -E "tensors, worst|passed|failed|FAILED|Error" > "$OUT/A_train_shipped_db.log"
# B: per-convolution audit, shipped db
timeout 600 python tools/miopen_conv_audit.py shapehd 2 > "$OUT/B_audit_shapehd_shipped.log" 2>&1
# C: same test + audit, Winograd excluded
MIOPEN_DEBUG_CONV_WINOGRAD=0 timeout 600 python -m pytest tests/test_gpu_train.py -q -m gpu -k shapehd 2>&1 | grep -E "tensors, worst|passed|failed|FAILED|Error" > "$OUT/C_train_nowino.log"
MIOPEN_DEBUG_CONV_WINOGRAD=0 timeout 600 python tools/miopen_conv_audit.py shapehd 2 > "$OUT/D_audit_shapehd_nowino.log" 2>&1
# E: MIOpen's own log of the solver picked for the suspicious weight gradient (512x8x8 3x3, batch 2)
MIOPEN_LOG_LEVEL=6 timeout 300 python - > "$OUT/E_solver_log.txt" 2>&1 <<'PY'
import sys, os
sys.path[:0] = [os.getcwd()]
import miopen_cache; miopen_cache.use()
import torch
m = torch.nn.Conv2d(512, 512, 3, 1, 1, bias=False).cuda()
x = torch.randn(2, 512, 8, 8, device="cuda", requires_grad=True)
m(x).backward(torch.randn(2, 512, 8, 8, device="cuda"))
torch.cuda.synchronize()
PY
grep -E "Chosen|solver|Solution|algo" "$OUT/E_solver_log.txt" | head -60 > "$OUT/E_solver_log_short.txt"; rm -f "$OUT/E_solver_log.txt"
# F: kernel traces of the train steps (item 6)
cd /tmp
export MIOPEN_USER_DB_PATH=$MI/db MIOPEN_CUSTOM_CACHE_DIR=$MI/cache
for cfg in "shapehd 8" "genre 4" "wgangp 8"; do
  set -- $cfg
  timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$1" -o t -- python "$ROOT/genre-shapehd_amd/train.py" --config $1 --batch $2 --steps 8 > "$OUT/F_train_$1.log" 2>&1
  DB=$(ls "$OUT"/prof_$1/t_results.db "$OUT"/prof_$1/*/t_results.db 2>/dev/null | head -1)
  python "$ROOT/profiles/summarize_rocpd.py" "$DB" > "$OUT/F_train_${1}_kernel_stats.txt" 2>&1
  rm -rf "$OUT/prof_$1"
done
cd "$ROOT"
# keep what MIOpen compiled / found in this call
tar czf "$OUT/miopen_after.tgz" -C "$ROOT/genre-shapehd_amd" .miopen
cat "$OUT"/A_*.log "$OUT"/C_*.log; tail -3 "$OUT"/B_*.log; grep -- "<--" "$OUT"/B_*.log "$OUT"/D_*.log | head -40
