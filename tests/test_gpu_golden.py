"""GPU: the reference's own unit-test vectors through the product (host layer → C ABI → HIP kernels), via both the
typed methods and the `_dyn` functions — the same runner tests/test_oracle_golden.py uses for the oracle."""
import os

import numpy as np
import pytest

import golden_runner as G

pytestmark = pytest.mark.gpu

VECTORS = G.load("reference_vectors.json")
HAND = G.load("reference_handwritten.json")


@pytest.mark.parametrize("rec", VECTORS, ids=[r["name"] + "@" + r["ref"].split("/")[1] for r in VECTORS])
def test_reference_vector(ag, rec):
    G.run_vector(ag, rec)


@pytest.mark.parametrize("rec", [r for r in HAND if r["kind"] not in ("builder_set_bit", "builder_new_set")],
                         ids=lambda r: r["name"])
def test_reference_handwritten(ag, rec):
    out = G.run_handwritten(ag, rec)
    if rec["kind"] == "from_optional_and_merge":
        a, b = out
        assert list(a.null_buffer.raw_values()) == rec["a_validity"]
        assert list(b.null_buffer.raw_values()) == rec["b_validity"]
        merged = ag.NullBitBufferGpu.merge_null_bit_buffer(b.null_buffer, a.null_buffer)
        assert list(merged.raw_values()) == rec["merged_validity"]


def test_unsupported_pairs_raise_like_the_reference_panics(ag):
    dev = ag.GPU_DEVICE()
    f = ag.Float32ArrayGPU.from_slice([1.0, 2.0], dev)
    i = ag.Int32ArrayGPU.from_slice([1, 2], dev)
    u8 = ag.UInt8ArrayGPU.from_slice([1, 2], dev)
    with pytest.raises(ag.OperationNotSupported):
        ag.add_dyn(f, i)                      # mixed types
    with pytest.raises(ag.OperationNotSupported):
        ag.sub_array_dyn(i, i)                # sub is f32-only in the reference's table
    with pytest.raises(ag.OperationNotSupported):
        ag.neg_dyn(i)
    with pytest.raises(ag.OperationNotSupported):
        ag.sqrt_dyn(i)
    with pytest.raises(ag.CastingNotSupported):
        ag.cast_dyn(ag.UInt32ArrayGPU.from_slice([1, 2], dev), ag.ArrowType.Float32Type)  # u32 → f32 is not in the table
    with pytest.raises(ag.CastingNotSupported):
        ag.Int32ArrayGPU.try_from(f)
    with pytest.raises(ag.OperationNotSupported):
        ag.take_dyn(u8, ag.UInt32ArrayGPU.from_slice([0], dev))
    with pytest.raises(ag.ArrowErrorGPU):
        f.take(ag.UInt32ArrayGPU.from_slice([5], dev))  # out of range: HIP has no robust buffer access


def test_examples_simple_py_runs():
    """examples/simple.py — the reference's examples/simple.rs in the Python host layer."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "simple.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK on" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
