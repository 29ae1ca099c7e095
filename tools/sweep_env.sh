#!/bin/bash
# Run on the GPU box: bench lines (steps/s, ms/step) for a list of "VAR=value[,VAR=value...]" settings x workloads.
# usage: tools/sweep_env.sh "c2 c3" "PDMPC_FR_ROUND=384" "PDMPC_FR_ROUND=512,PDMPC_FR_RAMP=2" ...
set -u
cd "$GRAFT_REPO_ROOT"
CFGS=$1; shift
for setting in "$@"; do
  for c in $CFGS; do
    out=$(env $(echo "$setting" | tr ',' ' ') timeout 300 python bench.py --workload $c --steps 200 --warmup 20 --no-cpu-baseline --no-host-inclusive --no-scaling-reference 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), round(d['p50_latency_ms'],2), round(d['p99_latency_ms'],2), d['counters'].get('nodes_processed_per_step'), d['counters'].get('rounds_per_step'))")
    echo "$setting $c $out"
  done
done
