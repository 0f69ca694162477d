#!/bin/bash
# One command for the evidence bench.py quotes (run on the MI355X box from the repo root, e.g.
#   gpurun --timeout 900 -- 'bash profiles/collect_pmc.sh r02a'):
#   gpurun_out/<tag>/kernel_stats.txt     rocprofv3 --kernel-trace --stats of the default bench run
#   gpurun_out/<tag>/pmc_hbm_traffic.txt  FETCH_SIZE / WRITE_SIZE per kernel, separate passes over profiles/pmc_targets.py
#   gpurun_out/<tag>/pmc_hbm_traffic.json the same, stamped with the kernel-source hash: copy to profiles/<tag>_pmc_hbm_traffic.json
#   gpurun_out/<tag>/bench.json           the bench line of the same build
set -u
TAG=${1:-r02}
B=${2:-32}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o bench -- python "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-m1 > "$OUT/bench_prof.json" 2> "$OUT/bench_prof.err"
python "$ROOT/profiles/summarize_rocpd.py" "$OUT"/prof/bench_results.db > "$OUT/kernel_stats.txt" 2>&1 || python "$ROOT/profiles/summarize_rocpd.py" $(ls "$OUT"/prof/*/*_results.db | head -1) > "$OUT/kernel_stats.txt" 2>&1
rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_f" -o fetch -- python "$ROOT/profiles/pmc_targets.py" "$B" > /dev/null 2> "$OUT/pmc_f.err"
rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_w" -o write -- python "$ROOT/profiles/pmc_targets.py" "$B" > /dev/null 2> "$OUT/pmc_w.err"
F=$(ls "$OUT"/pmc_f/fetch_results.db "$OUT"/pmc_f/*/fetch_results.db 2>/dev/null | head -1)
W=$(ls "$OUT"/pmc_w/write_results.db "$OUT"/pmc_w/*/write_results.db 2>/dev/null | head -1)
python "$ROOT/profiles/pmc_traffic_table.py" "$F" "$W" "$OUT/pmc_hbm_traffic.json" "$B" > "$OUT/pmc_hbm_traffic.txt" 2>&1
cd "$ROOT" && python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 600 "$OUT/bench.json"; echo; head -25 "$OUT/kernel_stats.txt"; cat "$OUT/pmc_hbm_traffic.txt"
rm -rf "$OUT"/pmc_f "$OUT"/pmc_w "$OUT"/prof
