#!/bin/bash
# round 4, FINAL GPU call (after the image-minor camera kernel became opt-in): full GPU suite as the driver runs it + smoke() +
# the default bench line, on the tree and MIOpen cache that ship
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04k; mkdir -p "$OUT"
export TMPDIR=/tmp
T0=$(date +%s)
timeout 600 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -30 | cut -c1-400 > "$OUT/pytest_gpu.log"
echo "pytest wall: $(( $(date +%s) - T0 )) s" >> "$OUT/pytest_gpu.log"
python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
timeout 200 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -5 "$OUT/pytest_gpu.log"; tail -2 "$OUT/smoke.log"; tail -c 600 "$OUT/bench.json"
