#!/bin/bash
# first contact of a kernel change with the GPU: the bulk kernel on batches of independent problems, then the step-level tests
cd "$GRAFT_REPO_ROOT"
export PDMPC_SPIN_LIMIT=${PDMPC_SPIN_LIMIT:-200000}
mkdir -p gpurun_out
for args in "1 4 6" "1 24 6" "2 24 8" "3 64 8"; do
  echo "== bk_debug $args"
  timeout 120 python tools/bk_debug.py $args 2>&1 | tail -30
done
echo "== parity tests"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -15
echo "== step tests"
timeout 900 python -m pytest tests/test_gpu_step.py -x -q 2>&1 | tail -15
