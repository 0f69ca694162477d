#!/bin/bash
# round 4, FINAL GPU call: the full GPU suite exactly as the driver runs it + smoke() on the tree and MIOpen cache that ship,
# then the evidence of the same tree (collect_pmc: kernel traces, PMC traffic, SQ counters, bench line)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04j; mkdir -p "$OUT"
export TMPDIR=/tmp
T0=$(date +%s)
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -40 | cut -c1-400 > "$OUT/pytest_gpu.log"
echo "pytest wall: $(( $(date +%s) - T0 )) s" >> "$OUT/pytest_gpu.log"
python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
bash profiles/collect_pmc.sh r04j > "$OUT/collect.log" 2>&1
tar czf "$OUT/miopen_after.tgz" -C "$ROOT/genre-shapehd_amd" .miopen
tail -6 "$OUT/pytest_gpu.log"; tail -2 "$OUT/smoke.log"; tail -c 800 "$OUT/bench.json"; grep -E "bm_combine_bwd|bm_zero_shared|bm_scatter" "$OUT/kernel_stats_soft.txt" | cut -c1-150
