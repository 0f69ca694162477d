#!/bin/bash
# round 4, GPU call 6: full GPU suite (as the driver runs it), train knobs, profiles + bench line (collect_pmc), smoke; the MIOpen
# cache after all of it is taken home so that the FINAL call runs on exactly the artefacts that ship
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04f; mkdir -p "$OUT"
export TMPDIR=/tmp
T0=$(date +%s)
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -40 > "$OUT/pytest_gpu.log"
echo "pytest wall: $(( $(date +%s) - T0 )) s" >> "$OUT/pytest_gpu.log"
python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
for m in default benchmark channels_last; do timeout 900 python tools/train_knobs.py $m 2>&1 | grep -E "KNOB|Error" >> "$OUT/knobs.log"; done
bash profiles/collect_pmc.sh r04f > "$OUT/collect.log" 2>&1
tar czf "$OUT/miopen_after.tgz" -C "$ROOT/genre-shapehd_amd" .miopen
tail -5 "$OUT/pytest_gpu.log"; cat "$OUT/smoke.log" | tail -2; cat "$OUT/knobs.log"; tail -40 "$OUT/collect.log" | cut -c1-250
