# A/B of PDMPC_TUNING settings on the C2 replay (bench.py without the CPU and host-inclusive legs): steps/s, p50, p99 per setting, twice
for rep in 1 2; do
for t in "$@"; do
  PDMPC_TUNING="$t" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-inclusive --no-scaling-reference ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.readline());print('%-28s' % '$t', round(d['value'],1), round(d['p50_latency_ms'],3), round(d['p99_latency_ms'],3))"
done
done
