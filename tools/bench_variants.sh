#!/bin/bash
# Run on the GPU box: bench lines (steps/s, ms/step, parity) of library variants under build/variants/.  usage: bench_variants.sh "c2 c3" name1 name2 ...
ws=$1; shift
for v in "$@"; do
  for w in $ws; do
    PDMPC_LIB=/root/repo/build/variants/libpdmpc_$v.so python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --no-scaling-reference --no-host-inclusive 2>/dev/null | grep metric > /tmp/b.json
    python - <<PY
import json
try:
    b = json.loads(open("/tmp/b.json").read())
    print("$v", "$w", round(b["value"], 1), round(b["ms_per_step"], 3), "p50", round(b["p50_latency_ms"], 3), "p99", round(b["p99_latency_ms"], 3), "replay_mismatches", b.get("replay_mismatches"))
except Exception as e:
    print("$v", "$w", "ERR", e)
PY
  done
done
