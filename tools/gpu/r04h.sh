#!/bin/bash
# round 4, GPU call 8: MIOpen's measured find for every network shape the bench / tests / train steps use (tools/warm_miopen.py
# with cudnn.benchmark), then the bench line under the resulting find-db
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04h; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1500 python tools/warm_miopen.py > "$OUT/warm.log" 2>&1
timeout 900 python bench.py --cpu-seconds 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
tar czf "$OUT/miopen_after.tgz" -C "$ROOT/genre-shapehd_amd" .miopen
tail -6 "$OUT/warm.log"; tail -c 1500 "$OUT/bench.json"
