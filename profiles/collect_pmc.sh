#!/bin/bash
# One command for the evidence bench.py quotes (run on the MI355X box from the repo root, e.g.
#   gpurun --timeout 900 -- 'bash profiles/collect_pmc.sh r02a'):
#   gpurun_out/<tag>/kernel_stats.txt     rocprofv3 --kernel-trace --stats of the default bench run
#   gpurun_out/<tag>/kernel_stats_phases.txt  the same for profiles/pmc_targets.py, renderer kernels split into @genre / @soft phases
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
rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o bench -- python "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-m1 --no-train > "$OUT/bench_prof.json" 2> "$OUT/bench_prof.err"
python "$ROOT/profiles/summarize_rocpd.py" "$OUT"/prof/bench_results.db > "$OUT/kernel_stats.txt" 2>&1 || python "$ROOT/profiles/summarize_rocpd.py" $(ls "$OUT"/prof/*/*_results.db | head -1) > "$OUT/kernel_stats.txt" 2>&1
# the headline `value` is the GenRe full-model forward: its own kernels, batch 1 and batch 8 (profiles/m1_target.py, eager launches)
for b in 1 8; do
rocprofv3 --kernel-trace --stats -d "$OUT/prof_m1_b$b" -o m1 -- python "$ROOT/profiles/m1_target.py" $b 10 > /dev/null 2> "$OUT/prof_m1_b$b.err"
python "$ROOT/profiles/summarize_rocpd.py" $(ls "$OUT"/prof_m1_b$b/m1_results.db "$OUT"/prof_m1_b$b/*/m1_results.db 2>/dev/null | head -1) > "$OUT/m1_b${b}_kernel_stats.txt" 2>&1
rm -rf "$OUT/prof_m1_b$b"
done
# kernel trace of pmc_targets.py alone: the renderers run there in two phases, GenRe's own volume (what the timed step renders and
# `roofline` is quoted on) then the soft volume (`roofline_soft`); min_us / max_us of a kernel's row are the two phases
rocprofv3 --kernel-trace --stats -d "$OUT/prof_t" -o targets -- python "$ROOT/profiles/pmc_targets.py" "$B" 16 > /dev/null 2> "$OUT/prof_t.err"
python "$ROOT/profiles/summarize_rocpd.py" --phases $(ls "$OUT"/prof_t/targets_results.db "$OUT"/prof_t/*/targets_results.db 2>/dev/null | head -1) > "$OUT/kernel_stats_phases.txt" 2>&1
rm -rf "$OUT/prof_t"
rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_f" -o fetch -- python "$ROOT/profiles/pmc_targets.py" "$B" > /dev/null 2> "$OUT/pmc_f.err"
rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_w" -o write -- python "$ROOT/profiles/pmc_targets.py" "$B" > /dev/null 2> "$OUT/pmc_w.err"
F=$(ls "$OUT"/pmc_f/fetch_results.db "$OUT"/pmc_f/*/fetch_results.db 2>/dev/null | head -1)
W=$(ls "$OUT"/pmc_w/write_results.db "$OUT"/pmc_w/*/write_results.db 2>/dev/null | head -1)
python "$ROOT/profiles/pmc_traffic_table.py" "$F" "$W" "$OUT/pmc_hbm_traffic.json" "$B" > "$OUT/pmc_hbm_traffic.txt" 2>&1
# SQ counters of the two batch-minor render kernels (8 SQ slots per pass on gfx950), no trace domains beside --pmc
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d "$OUT/pmc_s1" -o sq1 -- python "$ROOT/profiles/pmc_targets.py" "$B" > /dev/null 2> "$OUT/pmc_s1.err"
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU -d "$OUT/pmc_s2" -o sq2 -- python "$ROOT/profiles/pmc_targets.py" "$B" > /dev/null 2> "$OUT/pmc_s2.err"
S1=$(ls "$OUT"/pmc_s1/sq1_results.db "$OUT"/pmc_s1/*/sq1_results.db 2>/dev/null | head -1)
S2=$(ls "$OUT"/pmc_s2/sq2_results.db "$OUT"/pmc_s2/*/sq2_results.db 2>/dev/null | head -1)
python "$ROOT/profiles/pmc_sq_table.py" $S1 $S2 -- bm_scatter_kernel bm_sample_kernel bm_combine cam_brick seg_sample seg_combine seg_scatter cam_backward > "$OUT/sq_counters.txt" 2>&1
# the bench line of the same build reads the table just measured (roofline.traffic must not be null in a committed line)
cp "$OUT/pmc_hbm_traffic.json" "$ROOT/profiles/${TAG}_pmc_hbm_traffic.json"
cd "$ROOT" && T0=$(date +%s) && python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench.py wall time: $(( $(date +%s) - T0 )) s" >> "$OUT/bench.err"
if ! python - "$OUT/bench.json" <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")][-1]
sys.exit(0 if json.loads(line)["roofline"]["traffic"] is not None else 1)
PY
then mv "$OUT/bench.json" "$OUT/bench_REJECTED_traffic_null.json"; echo "collect_pmc.sh: roofline.traffic is null -- bench line rejected" >&2; fi
tail -c 600 "$OUT/bench.json"; echo; head -25 "$OUT/kernel_stats.txt"; cat "$OUT/pmc_hbm_traffic.txt"
cat "$OUT/sq_counters.txt"
rm -rf "$OUT"/pmc_f "$OUT"/pmc_w "$OUT"/pmc_s1 "$OUT"/pmc_s2 "$OUT"/prof
