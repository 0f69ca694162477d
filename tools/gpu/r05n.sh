#!/bin/bash
# round 5, last GPU action: the driver's round-end sequence (full GPU suite under -x, smoke) + the evidence run on the tree that ships
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05n; mkdir -p $OUT
T0=$(date +%s)
timeout 2400 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $? in $(( $(date +%s) - T0 )) s" >> $OUT/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?" >> $OUT/smoke.log
tail -4 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log
bash profiles/collect_pmc.sh r05n 32 > $OUT/collect.log 2>&1; tail -3 $OUT/collect.log
