#!/bin/bash
# Run on the GPU box: bench lines (value, replay mismatches, bad statuses) of the in-tree library against other builds of it.
#   tools/ab.sh "c2 c4" 3 tools/experiments/libX.so ...   (workloads, repetitions, libraries; "-" = the in-tree build)
W=${1:-c2}; N=${2:-3}; shift 2
for lib in - "$@"; do
  for w in $W; do
    for i in $(seq $N); do
      if [ "$lib" = "-" ]; then unset PDMPC_LIB; else export PDMPC_LIB=$PWD/$lib; fi
      python bench.py --workload $w --no-cpu-baseline --no-host-inclusive --no-scaling-reference 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.readline());print('$lib', '$w', round(d['value'],1), d.get('replay_mismatches'), d.get('bad_status_plans_in_timed_region'))"
    done
  done
done
