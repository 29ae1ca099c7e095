"""Build-time properties of the product's kernels (no GPU needed: hipcc cross-compiles gfx950 code objects)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "p-dmpc_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

SEARCH_KERNELS = ("pdmpc_bulk_kernel", "pdmpc_bulk_kernel_wide", "pdmpc_bulk_kernel_sat", "pdmpc_bulk_kernel_compact")


@pytest.mark.timeout(900)
def test_search_kernels_use_no_scratch_memory_and_spill_no_vgprs():
    """the instantiations of the graph search (InterX with one successor-mask word, InterX with any
    number, the separating-axis checker, and the InterX kernel built for two workgroups per CU) fit the register budget of their workgroups — sixteen wavefronts for the InterX kernels
    (four per SIMD: 128 VGPRs), twelve for the separating-axis kernel (168) — without a byte of scratch memory.  `make resources` compiles every kernel with the flags of the build (the inliner's basic-block limit raised, no
    machine LICM for the search) and prints the compiler's resource remarks."""
    if not os.path.exists(HIPCC) and shutil.which("hipcc") is None:
        pytest.skip("no hipcc")
    out = subprocess.run(["make", "-s", "-C", CSRC, "resources"], capture_output=True, text=True, check=True).stdout
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
    for kernel in SEARCH_KERNELS:
        assert kernel in seen, seen.keys()
        assert seen[kernel]["scratch"] == 0 and seen[kernel]["vgpr_spill"] == 0, (kernel, seen[kernel])
        assert seen[kernel]["vgprs"] <= (168 if kernel.endswith("_sat") else 128), (kernel, seen[kernel])


def test_no_legacy_search_kernels_are_shipped():
    """One product kernel: the pop-ordered kernel of round 1 and the one-node-per-wavefront kernel of rounds 2-3 are gone."""
    for name in ("serial_search.hpp", "search_kernel.hip", "frontier_kernel.hip", "blockmin_queue.hpp"):
        assert not os.path.exists(os.path.join(CSRC, name)), name
