#!/bin/bash
# the driver's round-end sequence on the tree as it is: full GPU suite under -x, smoke()
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05i; mkdir -p $OUT
T0=$(date +%s)
timeout 2400 python -m pytest tests/ -x -q -m gpu -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $? in $(( $(date +%s) - T0 )) s" >> $OUT/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?" >> $OUT/smoke.log
grep -E "passed|failed|FAILED|^E  |non-vacuous|rc " $OUT/pytest_gpu.log | tail -30; tail -3 $OUT/smoke.log
