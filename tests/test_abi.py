"""The C-ABI library loads and exports every symbol include/pdmpc.h declares (no compute calls: no GPU here)."""
import ctypes
import os
import re

import pytest

from pdmpc import abi, backend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "pdmpc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pdmpc_[a-z_]+)\s*\(", text)))


def test_header_declares_the_documented_entry_points():
    names = declared_functions()
    for must in ("pdmpc_create", "pdmpc_upload_mpa", "pdmpc_plan_batch", "pdmpc_get_last_stats", "pdmpc_destroy"):
        assert must in names


def test_library_exports_every_declared_symbol():
    lib = backend.load_library()
    for name in declared_functions():
        assert hasattr(lib, name), "libpdmpc_hip.so does not export %s" % name
    assert sorted(backend.EXPORTS) == declared_functions()
    assert b"gfx950" in lib.pdmpc_version()


def test_struct_layouts_match_the_header():
    # sizes the C compiler produces for include/pdmpc.h (natural alignment, no packing)
    assert ctypes.sizeof(abi.Config) == 32
    assert ctypes.sizeof(abi.Maneuver) == 8 * 3 + 8 + 3 * 2 * abi.VMAX * 8
    assert ctypes.sizeof(abi.PolygonSet) == 32
    assert ctypes.sizeof(abi.VehicleOut) == abi.VEHICLE_OUT_DTYPE.itemsize
    assert abi.VEHICLE_OUT_DTYPE.fields["y_predicted"][1] % 8 == 0


def test_no_device_fails_loudly_without_cpu_fallback():
    """On a box without a GPU the backend must refuse to work instead of silently computing on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from pdmpc.config import Config

    with pytest.raises(backend.BackendError) as e:
        backend.Handle(Config(Hp=5))
    assert "no CPU fallback" in str(e.value) or "no HIP device" in str(e.value)


def test_product_code_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under p-dmpc_amd/ may import, link, load or call it."""
    pkg = os.path.join(ROOT, "p-dmpc_amd")
    forbidden = re.compile(r"(from\s+oracle|import\s+oracle|libpdmpc_oracle|oracle_[a-z_]+\s*\(|oracle/|pdmpc_oracle\.cpp)")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".m")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                m = forbidden.search(text)
                assert m is None, "%s references the oracle: %r" % (os.path.join(dirpath, f), m.group(0))
