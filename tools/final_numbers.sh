#!/bin/bash
# Run on the GPU box: the round's profiles (rocprofv3 passes of C2, C4, C5) and the full bench lines of every workload.
for w in c2 c5 c4; do WORKLOAD=$w bash tools/collect_profiles.sh > gpurun_out/collect_$w.log 2>&1; done
python bench.py --steps 200 --warmup 20 2>/dev/null | grep metric > gpurun_out/r03_bench_c2.json
for w in c3 c4 c5; do python bench.py --workload $w 2>/dev/null | grep metric > gpurun_out/r03_bench_$w.json; done
PDMPC_DEBUG_TAIL=1 python tools/fr_step_profile.py > gpurun_out/r03_step_profile.txt 2>&1
ls -la gpurun_out | tail -20
