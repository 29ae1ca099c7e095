#!/bin/bash
# Run on the GPU box: the round's full bench lines of every workload, the step profiles and the kernels' resource usage, all under
# gpurun_out/ (the rocprofv3 passes are tools/collect_profiles.sh: WORKLOAD=c2|c3|c4|c5 ROUND=r06, EXTRA / NAME for the C2 grid).
R=${ROUND:-r06}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python bench.py --steps 200 --warmup 20 2>/dev/null | grep metric > gpurun_out/${R}_bench_c2.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep metric > gpurun_out/${R}_bench_c2_driver_window.json
for w in c3 c4 c5; do python bench.py --workload $w 2>/dev/null | grep metric > gpurun_out/${R}_bench_$w.json; done
# SURVEY.md 8(d)'s grid for C2 (eval_experiments.m:25-41): single_speed and triple_speed, seeds 1..3, each against the oracle
python bench.py --steps 200 --warmup 20 --mpa triple_speed 2>/dev/null | grep metric > gpurun_out/${R}_bench_c2triple.json
for sd in 2 3; do python bench.py --steps 200 --warmup 20 --seed $sd 2>/dev/null | grep metric > gpurun_out/${R}_bench_c2seed$sd.json; done
for sd in 2 3; do python bench.py --steps 200 --warmup 20 --seed $sd --mpa triple_speed --no-host-inclusive 2>/dev/null | grep metric > gpurun_out/${R}_bench_c2triple_seed$sd.json; done
# the group path with four logical ranks on this one GPU (csrc/group.cpp: everything a group of four devices executes but the transport)
for w in c3 c4 c5; do PDMPC_FORCE_GROUP=1 PDMPC_GROUP_LOGICAL=4 python bench.py --workload $w --no-cpu-baseline --no-host-inclusive 2>/dev/null | grep metric > gpurun_out/${R}_bench_${w}_group4.json; done
PDMPC_FORCE_GROUP=1 PDMPC_GROUP_LOGICAL=4 python bench.py --workload c4 --shard levels --no-cpu-baseline --no-host-inclusive 2>/dev/null | grep metric > gpurun_out/${R}_bench_c4_group4_levels.json
PROFILE_CHAIN=1 PROFILE_TOP=6 python tools/fr_step_profile.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_step_profile.txt
PROFILE_PASSES=1 PROFILE_TOP=2 python tools/fr_step_profile.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_step_profile_passes.txt
PROFILE_SEATS=1 PROFILE_TOP=6 python tools/fr_step_profile.py c4 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_step_profile_c4.txt
PROFILE_SEATS=1 PROFILE_TOP=4 python tools/fr_step_profile.py c3 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_step_profile_c3.txt
PROFILE_TOP=6 python tools/fr_step_profile.py c5 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_step_profile_c5.txt
make -C p-dmpc_amd/csrc resources > gpurun_out/${R}_resource_usage.txt 2>&1
python tools/print_bench_lines.py gpurun_out/${R}_bench_*.json
