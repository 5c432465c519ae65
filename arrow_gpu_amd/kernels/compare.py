"""arrow_gpu_compare: gt / gteq / lt / lteq / eq → BooleanArrayGPU, element-wise min / max.

Mirror of crates/compare/src/lib.rs (traits Compare, MinMax :41-83; launch macros :85-140; dyn tables :174-334).
Where the reference issues two steps (compare dispatch, then the validity AND submitted separately —
null_bit_buffer.rs:206-243), this calls the fused `agpu_compare_validity` kernel: one pass over the data plus the two
validity bitmaps.
"""
from __future__ import annotations

from .. import _capi as capi
from ..array import (BooleanArrayGPU, Date32ArrayGPU, Float32ArrayGPU, Int8ArrayGPU, Int16ArrayGPU, Int32ArrayGPU,
                     NullBitBufferGpu, UInt8ArrayGPU, UInt16ArrayGPU, UInt32ArrayGPU, bitmap_bytes)
from ..gpu_utils import ArrowComputePipeline
from ._ops import _finish_record, _recordable, array_op, check_same_len, dyn_binary, impl, vp

_ALL = (Float32ArrayGPU, UInt32ArrayGPU, UInt16ArrayGPU, UInt8ArrayGPU, Int32ArrayGPU, Int16ArrayGPU, Int8ArrayGPU,
        Date32ArrayGPU)


def _cmp_op(op: int):
    def fn(self, operand, pipeline: ArrowComputePipeline) -> BooleanArrayGPU:
        check_same_len(self, operand, "compare")
        dev = self.gpu_device
        n = self.len
        out = dev.create_empty_buffer(max(bitmap_bytes(n), 8))
        va, vb = self.null_buffer, operand.null_buffer
        nulls = None
        if _recordable(pipeline, 2, capi.OP_ADD, self.DTYPE, operand.ITEM_SIZE == self.ITEM_SIZE):
            # fusing pipeline: the compare is only recorded — it may end a fused chain (agpu_fused_chain_compare) whose
            # value is then never stored; validity is bitmap work and does not wait for the recorded value kernels
            node = pipeline.record_elementwise(3, op, self.DTYPE, self.data, operand.data, out, n)
            result = BooleanArrayGPU(out, dev, n, NullBitBufferGpu.merge_null_bit_buffer_op(va, vb, pipeline))
            return _finish_record(node, result)
        if va is None and vb is None:
            capi.call("agpu_compare", pipeline._handle, op, self.DTYPE, vp(self.data), vp(operand.data), vp(out), n)
        else:
            outv = dev.create_empty_buffer(max(bitmap_bytes(n), 8))
            cnt = dev.create_empty_buffer(8)  # the result's null count, a by-product of the validity blocks of the launch
            capi.call("agpu_compare_validity_count", pipeline._handle, op, self.DTYPE, vp(self.data), vp(operand.data),
                      vp(va.bit_buffer) if va else None, vp(vb.bit_buffer) if vb else None, vp(out), vp(outv), n, vp(cnt))
            nulls = NullBitBufferGpu(outv, n, dev, cnt, False)
            pipeline.keep(va.bit_buffer if va else None, vb.bit_buffer if vb else None, outv, cnt)
        pipeline.keep(self.data, operand.data, out)
        return BooleanArrayGPU(out, dev, n, nulls)

    return fn


impl(_ALL, "gt", _cmp_op(capi.CMP_GT))
impl(_ALL, "gteq", _cmp_op(capi.CMP_GTEQ))
impl(_ALL, "lt", _cmp_op(capi.CMP_LT))
impl(_ALL, "lteq", _cmp_op(capi.CMP_LTEQ))
impl(_ALL, "eq", _cmp_op(capi.CMP_EQ))

# MinMax: the reference's u8/i8/i16 MIN_MAX_SHADER points at the cmp shader (compare/src/u8.rs:8-12) and is broken;
# here every type works.
impl(_ALL, "max", array_op(capi.OP_MAX))
impl(_ALL, "min", array_op(capi.OP_MIN))

gt_dyn, gt_op_dyn = dyn_binary("gt_dyn", "gt_op", _ALL)
gteq_dyn, gteq_op_dyn = dyn_binary("gteq_dyn", "gteq_op", _ALL)
lt_dyn, lt_op_dyn = dyn_binary("lt_dyn", "lt_op", _ALL)
lteq_dyn, lteq_op_dyn = dyn_binary("lteq_dyn", "lteq_op", _ALL)
eq_dyn, eq_op_dyn = dyn_binary("eq_dyn", "eq_op", _ALL)
max_dyn, max_op_dyn = dyn_binary("max_dyn", "max_op", _ALL)
min_dyn, min_op_dyn = dyn_binary("min_dyn", "min_op", _ALL)

__all__ = [n for n in dir() if n.endswith("_dyn")]
