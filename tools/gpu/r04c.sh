#!/bin/bash
# round 4, GPU call 3: the failing tests of call 2 with full logs, the new defaults (cam_brick pixel screen + nontemporal stores,
# calc_prob plain loads at cache-resident sizes, GEMM weight gradients at >= 64^3), the restructured bench line
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04c; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1500 python -m pytest "tests/test_gpu_render.py::test_batch_minor_backward_skips_what_the_clamp_blocks" tests/test_gpu_thin_conv.py tests/test_gpu_z_train.py tests/test_gpu_cam_bp.py tests/test_gpu_calc_prob.py -q -m gpu --tb=short -s 2>&1 | cut -c1-600 > "$OUT/A_pytest.log"
timeout 300 python tools/ab_round4.py --worker m2 default > "$OUT/B_m2.log" 2>&1
timeout 1500 python bench.py --steps 10 --train-steps 4 --train-configs all --cpu-seconds 5 > "$OUT/C_bench.json" 2> "$OUT/C_bench.err"
tar czf "$OUT/miopen_after.tgz" -C "$ROOT/genre-shapehd_amd" .miopen
grep -E "passed|failed|FAILED|^E  " "$OUT/A_pytest.log" | head -60; cat "$OUT/B_m2.log" | grep AB4; tail -c 3000 "$OUT/C_bench.json"; tail -5 "$OUT/C_bench.err"
