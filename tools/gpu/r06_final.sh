#!/bin/bash
# the round's last GPU action: the whole GPU suite + smoke on the tree that ships
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_final; mkdir -p $OUT
( timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -8; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) | tee $OUT/final_pytest_gpu.log
