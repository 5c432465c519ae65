"""CPU: the C-ABI shared library loads, exports EVERY function include/arrow_gpu.h declares (and the ctypes table
matches), and fails loudly without a GPU — no compute calls here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "arrow_gpu.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(agpu_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_documented_surface():
    fns = declared_functions()
    assert len(fns) >= 55
    for must in ("agpu_device_create", "agpu_pipeline_create", "agpu_binary", "agpu_scalar", "agpu_unary", "agpu_cast",
                 "agpu_compare", "agpu_compare_validity", "agpu_bitmap_binary", "agpu_bitmap_popcount", "agpu_reduce",
                 "agpu_take", "agpu_put", "agpu_merge", "agpu_broadcast", "agpu_launch_by_name"):
        assert must in fns


def test_library_exports_every_declared_symbol():
    from arrow_gpu_amd import _capi as capi

    lib = capi.lib()
    missing = [f for f in declared_functions() if not hasattr(lib, f)]
    assert not missing, f"declared in include/arrow_gpu.h but not exported: {missing}"


def test_ctypes_table_covers_header():
    from arrow_gpu_amd import _capi as capi

    declared = set(declared_functions())
    bound = set(capi.SIGNATURES)
    assert declared - bound == set(), f"unbound: {sorted(declared - bound)}"
    assert bound - declared == set(), f"bound but undeclared: {sorted(bound - declared)}"


def test_misc_queries_work_without_gpu():
    from arrow_gpu_amd import _capi as capi

    lib = capi.lib()
    assert lib.agpu_abi_version() == 2
    assert b"gfx950" in lib.agpu_build_info()
    assert lib.agpu_dtype_size(capi.F32) == 4 and lib.agpu_dtype_size(capi.I16) == 2 and lib.agpu_dtype_size(capi.BOOL) == 0
    assert lib.agpu_bitmap_bytes(0) == 0 and lib.agpu_bitmap_bytes(1) == 8 and lib.agpu_bitmap_bytes(65) == 16
    v = C.c_int64(-1)
    assert lib.agpu_get_tuning(b"stream_grid", C.byref(v)) == 0
    assert lib.agpu_set_tuning(b"no_such_key", 1) == capi.ERR_ARG


def test_no_cpu_fallback():
    """Without a device the product raises; it never computes on the host."""
    import torch

    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    import arrow_gpu_amd as ag

    with pytest.raises(ag.ArrowErrorGPU) as ei:
        ag.GpuDevice(0)
    assert ei.value.status == 5  # AGPU_ERR_NO_DEVICE


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "arrow_gpu_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dp, fn)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), f"{fn} imports the oracle"
                assert "liboracle" not in text and "agpu_oracle" not in text.replace("oracle/agpu_oracle.c", ""), fn
