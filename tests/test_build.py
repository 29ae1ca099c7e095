"""Build-time properties of the product's kernels (no GPU needed: hipcc cross-compiles gfx950 code objects)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "p-dmpc_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.timeout(600)
@pytest.mark.parametrize("source,kernels", [
    ("bulk_kernel.hip", ("pdmpc_bulk_kernel",)),
    ("frontier_kernel.hip", ("pdmpc_frontier_kernel", "pdmpc_frontier_kernel_sat", "pdmpc_frontier_kernel_wide", "pdmpc_frontier_kernel_sat_wide", "pdmpc_helper_kernel")),
])
def test_round_based_kernels_use_no_scratch_memory_and_spill_no_vgprs(source, kernels):
    """the product's search kernels (bulk: InterX; frontier: SAT and the tie fallback) and their helper kernels
    fit their register budget (168 VGPRs at twelve wavefronts per workgroup) without a byte of scratch memory.  The AMDGPU inliner gives up on functions of more than
    1100 basic blocks (the Makefile raises that limit); a search function left out of line takes the search context through the stack,
    so the build is checked."""
    if not os.path.exists(HIPCC) and shutil.which("hipcc") is None:
        pytest.skip("no hipcc")
    cmd = [HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           "-mllvm", "-amdgpu-inline-max-bb=10000",  # (as csrc/Makefile)
           *(["-mllvm", "-disable-machine-licm"] if source == "bulk_kernel.hip" else []),  # (BULK_FLAGS of csrc/Makefile)
           "-I" + os.path.join(ROOT, "include"), "-c", "-Rpass-analysis=kernel-resource-usage", "-o", os.devnull, os.path.join(CSRC, source)]
    out = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    seen = {}
    name = None
    for line in out.splitlines():
        m = re.search(r"Function Name: (\w+)", line)
        if m:
            name = m.group(1)
            seen[name] = {}
        for key, pat in (("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("vgpr_spill", r"VGPRs Spill: (\d+)"), ("vgprs", r" VGPRs: (\d+)")):
            m = re.search(pat, line)
            if m and name:
                seen[name][key] = int(m.group(1))
    for kernel in kernels:
        assert kernel in seen, seen.keys()
        assert seen[kernel]["scratch"] == 0 and seen[kernel]["vgpr_spill"] == 0, (kernel, seen[kernel])
        assert seen[kernel]["vgprs"] <= 168, (kernel, seen[kernel])
