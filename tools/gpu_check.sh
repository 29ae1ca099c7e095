#!/bin/bash
# Run on the GPU box (through gpurun): the GPU test suite, then one bench line per configuration (compact print).
# usage: tools/gpu_check.sh TAG [configs...]
set -u
cd "$GRAFT_REPO_ROOT"
TAG=${1:-chk}; shift || true
CFGS=${*:-c2 c3 c5}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for c in $CFGS; do
  timeout 600 python bench.py --workload $c --steps 200 --warmup 20 --no-cpu-baseline --no-host-inclusive > gpurun_out/${TAG}_bench_$c.json 2> gpurun_out/${TAG}_bench_$c.err
  python - "$c" gpurun_out/${TAG}_bench_$c.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d["value"], 1), round(d["ms_per_step"], 3), round(d.get("p50_latency_ms", 0), 2), round(d.get("p99_latency_ms", 0), 2))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
done
