#!/bin/bash
# round 5, the last GPU action: zero paths of the two backwards, then the round-end sequence + the evidence run on the tree that ships
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05t; mkdir -p $OUT
T0=$(date +%s)
timeout 2400 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc $rc in $(( $(date +%s) - T0 )) s" >> $OUT/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?" >> $OUT/smoke.log
tail -4 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log
if [ $rc -ne 0 ]; then exit 1; fi
bash profiles/collect_pmc.sh r05t 32 > $OUT/collect.log 2>&1; tail -3 $OUT/collect.log
