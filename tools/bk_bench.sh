#!/bin/bash
# bench lines of the workloads given (compact), for the kernel selected by PDMPC_KERNEL (default: bulk)
cd "$GRAFT_REPO_ROOT"
TAG=${1:-bk}; shift || true
CFGS=${*:-c2 c3 c5 c4}
mkdir -p gpurun_out
for c in $CFGS; do
  timeout 900 python bench.py --workload $c --steps 200 --warmup 20 --no-cpu-baseline --no-host-inclusive > gpurun_out/${TAG}_bench_$c.json 2> gpurun_out/${TAG}_bench_$c.err
  python - "$c" gpurun_out/${TAG}_bench_$c.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    c = d.get("counters", {})
    print(sys.argv[1], "steps/s", round(d["value"], 1), "ms", round(d["ms_per_step"], 3), "p50", round(d.get("p50_latency_ms", 0), 2), "p99", round(d.get("p99_latency_ms", 0), 2),
          "replay_mm", d.get("replay_mismatches"), "bad", d.get("bad_status_plans_in_timed_region"), "proc/step", c.get("nodes_processed_per_step"), "pops/step", c.get("nodes_popped_per_step"), "rounds/step", c.get("rounds_per_step"))
except Exception as e:
    print(sys.argv[1], "failed", e)
    print(open(sys.argv[2].replace(".json", ".err")).read()[-1500:])
PY
done
