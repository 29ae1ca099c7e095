#!/bin/bash
# Run on the GPU box: the round's full bench lines of every workload, the step profiles and the kernels' resource usage, all under
# gpurun_out/ (the rocprofv3 passes are tools/collect_profiles.sh: WORKLOAD=c2|c4|c5 ROUND=r05).
R=${ROUND:-r05}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python bench.py --steps 200 --warmup 20 2>/dev/null | grep metric > gpurun_out/${R}_bench_c2.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep metric > gpurun_out/${R}_bench_c2_driver_window.json
for w in c3 c4 c5; do python bench.py --workload $w 2>/dev/null | grep metric > gpurun_out/${R}_bench_$w.json; done
PROFILE_CHAIN=1 PROFILE_TOP=6 python tools/fr_step_profile.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_step_profile.txt
PROFILE_PASSES=1 PROFILE_TOP=2 python tools/fr_step_profile.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_step_profile_passes.txt
PROFILE_SEATS=1 PROFILE_TOP=6 python tools/fr_step_profile.py c4 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_step_profile_c4.txt
PROFILE_SEATS=1 PROFILE_TOP=4 python tools/fr_step_profile.py c3 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_step_profile_c3.txt
PROFILE_TOP=6 python tools/fr_step_profile.py c5 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_step_profile_c5.txt
make -C p-dmpc_amd/csrc resources > gpurun_out/${R}_resource_usage.txt 2>&1
python tools/print_bench_lines.py gpurun_out/${R}_bench_c2.json gpurun_out/${R}_bench_c2_driver_window.json gpurun_out/${R}_bench_c3.json gpurun_out/${R}_bench_c4.json gpurun_out/${R}_bench_c5.json
