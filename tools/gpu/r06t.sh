#!/bin/bash
# SQ counters of seg_scatter_kernel (soft volume, batch 32): where its wave cycles go
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06t; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d "$OUT/pmc_s1" -o sq1 -- python "$ROOT/tools/time_seg_bwd.py" 32 > "$OUT/pmc_s1.out" 2> "$OUT/pmc_s1.err"
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU -d "$OUT/pmc_s2" -o sq2 -- python "$ROOT/tools/time_seg_bwd.py" 32 > /dev/null 2> "$OUT/pmc_s2.err"
S1=$(ls "$OUT"/pmc_s1/sq1_results.db "$OUT"/pmc_s1/*/sq1_results.db 2>/dev/null | head -1)
S2=$(ls "$OUT"/pmc_s2/sq2_results.db "$OUT"/pmc_s2/*/sq2_results.db 2>/dev/null | head -1)
python "$ROOT/profiles/pmc_sq_table.py" $S1 $S2 -- seg_scatter seg_sample > "$OUT/sq_counters.txt" 2>&1
cat "$OUT/sq_counters.txt"
rm -rf "$OUT"/pmc_s1 "$OUT"/pmc_s2
grep -v simple_timer $OUT/pmc_s1.err | tail -5; cat $OUT/pmc_s1.out
