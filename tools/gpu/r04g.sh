#!/bin/bash
# round 4, GPU call 7: full GPU suite as the driver runs it + smoke + profiles / bench line (collect_pmc)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04g; mkdir -p "$OUT"
export TMPDIR=/tmp
T0=$(date +%s)
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -40 | cut -c1-400 > "$OUT/pytest_gpu.log"
echo "pytest wall: $(( $(date +%s) - T0 )) s" >> "$OUT/pytest_gpu.log"
python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
bash profiles/collect_pmc.sh r04g > "$OUT/collect.log" 2>&1
tar czf "$OUT/miopen_after.tgz" -C "$ROOT/genre-shapehd_amd" .miopen
tail -6 "$OUT/pytest_gpu.log"; tail -2 "$OUT/smoke.log"; tail -c 1200 "$OUT/bench.json"; cat "$OUT/pmc_hbm_traffic.txt" | head -8; head -12 "$OUT/kernel_stats_soft.txt" | cut -c1-150
