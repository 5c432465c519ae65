"""GPU: the last public names of SURVEY Appendix C (VERDICT r3 missing #4), one golden-style test per name, Python host; the C++ host's
counterparts run in tests/cpp/test_host_api.cpp.  Values follow the reference's own tests where it has any."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_uint32_create_broadcast_buffer_and_op(ag):
    """UInt32ArrayGPU::create_broadcast_buffer(_op) [crates/array/src/array/u32_gpu.rs:36-64]: a bare Buffer of `len` copies"""
    dev = ag.GPU_DEVICE()
    buf = ag.UInt32ArrayGPU.create_broadcast_buffer(7, 100, dev)
    assert buf.nbytes >= 400
    assert np.array_equal(dev.retrive_data(buf, 400).view(np.uint32), np.full(100, 7, np.uint32))
    p = ag.ArrowComputePipeline(dev, "bcast")
    buf2 = ag.UInt32ArrayGPU.create_broadcast_buffer_op(0xFFFFFFFF, 1_000_003, p)
    p.finish()
    got = dev.retrive_data(buf2, 4 * 1_000_003).view(np.uint32)
    assert got.min() == got.max() == 0xFFFFFFFF
    # the buffer is an ordinary one: it can become an array's data (what the reference does with it for index columns)
    arr = ag.UInt32ArrayGPU(buf, dev, 100, None)
    assert arr.raw_values().tolist() == [7] * 100 and int(arr.sum().raw_values()[0]) == 700


def test_scalar_array_is_what_get_raw_values_returns(ag):
    """ScalarArray [crates/array/src/utils/mod.rs:2-11] / ArrowArrayGPU::get_raw_values [array/src/array/mod.rs:145-157]"""
    dev = ag.GPU_DEVICE()
    cases = [(ag.Float32ArrayGPU, [1.5, -2.0, 0.0], "F32Vec"), (ag.UInt32ArrayGPU, [1, 2, 4000000000], "U32Vec"), (ag.UInt16ArrayGPU, [1, 65535], "U16Vec"),
             (ag.UInt8ArrayGPU, [0, 255], "U8Vec"), (ag.Int32ArrayGPU, [-1, 2], "I32Vec"), (ag.Int16ArrayGPU, [-32768, 7], "I16Vec"),
             (ag.Int8ArrayGPU, [-128, 127], "I8Vec"), (ag.Date32ArrayGPU, [19000, -5], "I32Vec"), (ag.BooleanArrayGPU, [True, False, True], "BOOLVec")]
    for cls, vals, kind in cases:
        raw = cls.from_slice(vals, dev).get_raw_values()
        assert isinstance(raw, ag.ScalarArray) and raw.kind == kind
        assert raw == getattr(ag.ScalarArray, kind)(vals)          # PartialEq: same variant, same values
        assert raw == vals and raw.tolist() == vals and len(raw) == len(vals)
    assert ag.ScalarArray.F32Vec([1, 2]) != ag.ScalarArray.U32Vec([1, 2])   # another variant is another value
    assert ag.ScalarArray.from_vec([1, 2], np.int16).kind == "I16Vec"
    with pytest.raises(ag.ArrowErrorGPU):
        ag.ScalarArray("F64Vec", [1.0])
    # the dyn functions hand back the same thing through the enum
    r = ag.add_dyn(ag.Float32ArrayGPU.from_slice([1.0, 2.0], dev), ag.Float32ArrayGPU.from_slice([10.0], dev))
    assert r.get_raw_values() == ag.ScalarArray.F32Vec([11.0, 12.0])


def test_operand_scalar_and_array(ag):
    """Operand { Scalar(ScalarValue), Array(ArrowArrayGPU) } [crates/array/src/kernels/mod.rs:19-24]"""
    dev = ag.GPU_DEVICE()
    a = ag.Int32ArrayGPU.from_slice([1, 2, 3], dev)
    s = ag.Operand.Scalar(ag.ScalarValue.I32(10))
    v = ag.Operand.Array(ag.Int32ArrayGPU.from_slice([100, 200, 300], dev))
    assert s.is_scalar() and not v.is_scalar()
    assert ag.add_dyn(a, s.as_array(dev)).get_raw_values() == [11, 12, 13]          # a 1-element array is the scalar form of *_dyn
    assert ag.add_dyn(a, v.as_array(dev)).get_raw_values() == [101, 202, 303]
    assert ag.Operand.Scalar(ag.ScalarValue.BOOL(True)).as_array(dev).raw_values().tolist() == [True]
    with pytest.raises(ag.ArrowErrorGPU):
        ag.Operand.Scalar(3)
    with pytest.raises(ag.ArrowErrorGPU):
        ag.Operand.Array([1, 2])


def test_apply_boolean_unary_function_is_the_boolean_cast(ag):
    """cast::apply_boolean_unary_function [crates/cast/src/boolean_cast.rs:8-55] with the reference's own arguments for
    BooleanArrayGPU → Float32ArrayGPU (:57-74): new_buffer_size = len · 4, output_item_size = 4, "cast_f32"; values of its test :83-90"""
    dev = ag.GPU_DEVICE()
    vals = [True, True, False, True, False]
    b = ag.BooleanArrayGPU.from_slice(vals, dev)
    p = ag.ArrowComputePipeline(dev, "bool-cast")
    out = ag.apply_boolean_unary_function(dev, b.data, b.len * 4, 4, "cast/boolean/cast_f32", "cast_f32", p)
    p.finish()
    assert dev.retrive_data(out, 20).view(np.float32).tolist() == [1.0, 1.0, 0.0, 1.0, 0.0]
    assert b.cast(ag.Float32ArrayGPU).raw_values().tolist() == [1.0, 1.0, 0.0, 1.0, 0.0]
    n = 100_003
    bits = np.random.default_rng(0).random(n) < 0.3
    big = ag.BooleanArrayGPU.from_slice(bits.tolist(), dev)
    out = ag.apply_boolean_unary_function(dev, big.data, n * 4, 4, "cast/boolean/cast_f32", "cast_f32", p)
    p.finish()
    assert np.array_equal(dev.retrive_data(out, 4 * n).view(np.float32), bits.astype(np.float32))


def test_gpu_device_from_adapter_is_the_ordinal(ag):
    """GpuDevice::from_adapter [gpu_device.rs:87-106]: the adapter of a ROCm node is the device ordinal"""
    dev = ag.GpuDevice.from_adapter(0)
    assert dev.ordinal == 0 and dev.name.startswith("gfx950")
    a = ag.Float32ArrayGPU.from_slice([1.0, 2.0], dev)
    assert a.add(a).raw_values().tolist() == [2.0, 4.0]
    with pytest.raises(ag.ArrowErrorGPU):
        ag.GpuDevice.from_adapter(99)
