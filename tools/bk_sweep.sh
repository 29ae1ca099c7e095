#!/bin/bash
# bench lines (steps/s, ms/step) for a list of "VAR=value[,VAR=value...]" settings x workloads.  usage: tools/bk_sweep.sh "c2 c3" "PDMPC_BK_ROUND=512" ...
cd "$GRAFT_REPO_ROOT"
CFGS=$1; shift
for setting in "$@"; do
  for c in $CFGS; do
    out=$(env $(echo "$setting" | tr ',' ' ') timeout 600 python bench.py --workload $c --steps 100 --warmup 10 --no-cpu-baseline --no-host-inclusive 2>/dev/null | tail -1)
    python - "$c" "$setting" "$out" <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[3])
    print(sys.argv[2], sys.argv[1], "steps/s", round(d["value"], 1), "ms", round(d["ms_per_step"], 3), "p50", round(d.get("p50_latency_ms", 0), 2), "p99", round(d.get("p99_latency_ms", 0), 2), "max", round(d.get("max_latency_ms", 0), 2), "mm", d.get("replay_mismatches"), "bad", d.get("bad_status_plans_in_timed_region"), "proc", round(d["counters"]["nodes_processed_per_step"]), "rounds", round(d["counters"]["rounds_per_step"]), "shared", round(d["counters"]["shared_rounds_per_step"]))
except Exception as e:
    print(sys.argv[2], sys.argv[1], "failed", e)
PY
  done
done
