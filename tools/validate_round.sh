#!/bin/bash
# One GPU-box pass over everything a kernel change must survive: the C3 closed loop N times (tools/dbg_loop.sh), the GPU test suite,
# and bench.py on every workload.  Results under gpurun_out/.   Usage: gpurun -- 'bash tools/validate_round.sh [N]'
n=${1:-60}
mkdir -p gpurun_out
bash tools/dbg_loop.sh $n "X=1" 2>/dev/null | grep -E "ok="
python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/gpu_tests.log | tail -3
python bench.py --steps 200 --warmup 20 2>/dev/null | grep metric > gpurun_out/b_c2.json
for w in c3 c4 c5; do python bench.py --workload $w --no-cpu-baseline --no-scaling-reference 2>/dev/null | grep metric > gpurun_out/b_$w.json; done
python - <<EOF
import json
for w in ("c2", "c3", "c4", "c5"):
    try:
        b = json.loads(open("gpurun_out/b_%s.json" % w).read())
        print(w, b["value"], b["ms_per_step"], b["roofline"]["frac"], b.get("parity_mismatches"), b.get("replay_mismatches"), b.get("value_host_inclusive"), b.get("value_run_optimizer_literal"))
    except Exception as e:
        print(w, "ERR", e)
EOF
