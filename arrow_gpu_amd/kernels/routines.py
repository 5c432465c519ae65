"""arrow_gpu_routines: Swizzle — merge (select by mask), take (gather), put (scatter).

Mirror of crates/routines/src/{lib,merge,take,put,bool}.rs.  Differences that are deliberate:
the taken validity bitmap has the INDEX length (the reference stores the source length, bool.rs:40-44); absent
validity bitmaps count as all-valid in `merge`; index ranges are checked inside the `take`/`put` kernels (HIP has no
robust buffer access: out-of-range reads yield 0 and writes are dropped, as WGSL would, and the pipeline's next `sync()`
raises ArrowErrorGPU ShapeError).
"""
from __future__ import annotations

from .. import _capi as capi
from .._capi import OperationNotSupported
from ..array import (ArrowArrayGPU, BooleanArrayGPU, Date32ArrayGPU, Float32ArrayGPU, Int8ArrayGPU, Int16ArrayGPU,
                     Int32ArrayGPU, NullBitBufferGpu, PrimitiveArrayGpu, UInt8ArrayGPU, UInt16ArrayGPU, UInt32ArrayGPU,
                     bitmap_bytes)
from ..gpu_utils import ArrowComputePipeline
from ._ops import check_same_len, impl, vp

_PRIMS = (Date32ArrayGPU, Int32ArrayGPU, Int16ArrayGPU, Int8ArrayGPU, UInt32ArrayGPU, UInt16ArrayGPU, UInt8ArrayGPU,
          Float32ArrayGPU)


def merge_null_buffers_op(op1, op2, mask: BooleanArrayGPU, pipeline: ArrowComputePipeline, n: int):
    """validity = ((v1 & m) | (v2 & ~m)) & v_mask in ONE kernel (the reference: up to 4 dispatches, merge.rs:17-86).
    All three absent → None."""
    vm = mask.null_buffer
    if op1 is None and op2 is None and vm is None:
        return None
    dev = mask.gpu_device
    out = dev.create_empty_buffer(max(bitmap_bytes(n), 8))
    capi.call("agpu_bitmap_merge_validity", pipeline._handle, vp(op1.bit_buffer) if op1 else None,
              vp(op2.bit_buffer) if op2 else None, vp(mask.data), vp(vm.bit_buffer) if vm else None, vp(out), n)
    pipeline.keep(op1.bit_buffer if op1 else None, op2.bit_buffer if op2 else None, mask.data,
                  vm.bit_buffer if vm else None, out)
    return NullBitBufferGpu(out, n, dev)


def take_null_buffer(null_buffer, indexes: UInt32ArrayGPU, pipeline: ArrowComputePipeline):
    """[ref: crates/routines/src/bool.rs:33-46]"""
    if null_buffer is None:
        return None
    dev = indexes.gpu_device
    out = dev.create_empty_buffer(max(bitmap_bytes(indexes.len), 8))
    capi.call("agpu_take_bits", pipeline._handle, vp(null_buffer.bit_buffer), null_buffer.len, vp(indexes.data), vp(out),
              indexes.len)
    pipeline.keep(null_buffer.bit_buffer, indexes.data, out)
    return NullBitBufferGpu(out, indexes.len, dev)


# ---- PrimitiveArrayGpu<T>  [crates/routines/src/lib.rs:81-171]
def _merge_op(self: PrimitiveArrayGpu, other, mask: BooleanArrayGPU, pipeline: ArrowComputePipeline):
    check_same_len(self, other, "merge")
    check_same_len(self, mask, "merge mask")
    dev = self.gpu_device
    out = dev.create_empty_buffer(max(self.len * self.ITEM_SIZE, 1))
    capi.call("agpu_merge", pipeline._handle, self.ITEM_SIZE, vp(self.data), vp(other.data), vp(mask.data), vp(out), self.len)
    pipeline.keep(self.data, other.data, mask.data, out)
    nulls = merge_null_buffers_op(self.null_buffer, other.null_buffer, mask, pipeline, self.len)
    return type(self)(out, dev, self.len, nulls)


def _take_op(self: PrimitiveArrayGpu, indexes: UInt32ArrayGPU, pipeline: ArrowComputePipeline):
    dev = self.gpu_device
    out = dev.create_empty_buffer(max(indexes.len * self.ITEM_SIZE, 1))
    if self.null_buffer is None:
        capi.call("agpu_take", pipeline._handle, self.ITEM_SIZE, vp(self.data), self.len, vp(indexes.data), vp(out), indexes.len)
        pipeline.keep(self.data, indexes.data, out)
        return type(self)(out, dev, indexes.len, None)
    # values and validity in ONE call: at bucketed sizes the validity bit travels with the value (agpu_take_validity) instead
    # of a second random pass over the index column [ref: take.rs:9-55 + bool.rs:33-46, two dispatches]
    outv = dev.create_empty_buffer(max(bitmap_bytes(indexes.len), 8))
    capi.call("agpu_take_validity", pipeline._handle, self.ITEM_SIZE, vp(self.data), self.len, vp(self.null_buffer.bit_buffer),
              vp(indexes.data), vp(out), vp(outv), indexes.len)
    pipeline.keep(self.data, self.null_buffer.bit_buffer, indexes.data, out, outv)
    return type(self)(out, dev, indexes.len, NullBitBufferGpu(outv, indexes.len, dev))


def _all_valid(dev, n: int, pipeline: ArrowComputePipeline) -> NullBitBufferGpu:
    out = dev.create_empty_buffer(max(bitmap_bytes(n), 8))
    capi.call("agpu_broadcast", pipeline._handle, capi.BOOL, 1, vp(out), n)
    pipeline.keep(out)
    return NullBitBufferGpu(out, n, dev)


def _put_validity(src, src_indexes, dst, dst_indexes, pipeline: ArrowComputePipeline) -> None:
    """Null-aware put (`todo!()` in the reference, routines/src/lib.rs:164-169): the validity bit travels
    with the value, dst.validity[dst_idx[i]] = src.validity[src_idx[i]].  An absent bitmap counts as all-valid; `dst`
    gains a validity bitmap when `src` has one.  Duplicate destination indices: unspecified winner, like the values."""
    if src.null_buffer is None and dst.null_buffer is None:
        return
    dev = dst.gpu_device
    sv = src.null_buffer if src.null_buffer is not None else _all_valid(dev, src.len, pipeline)
    if dst.null_buffer is None:
        dst.null_buffer = _all_valid(dev, dst.len, pipeline)
    capi.call("agpu_put_bits_bounded", pipeline._handle, vp(sv.bit_buffer), src.len, vp(src_indexes.data),
              vp(dst.null_buffer.bit_buffer), dst.len, vp(dst_indexes.data), src_indexes.len)
    pipeline.keep(sv.bit_buffer, dst.null_buffer.bit_buffer, src_indexes.data, dst_indexes.data)


def _put_op(self: PrimitiveArrayGpu, src_indexes: UInt32ArrayGPU, dst, dst_indexes: UInt32ArrayGPU,
            pipeline: ArrowComputePipeline) -> None:
    check_same_len(src_indexes, dst_indexes, "put indexes")
    capi.call("agpu_put_bounded", pipeline._handle, self.ITEM_SIZE, vp(self.data), self.len, vp(src_indexes.data),
              vp(dst.data), dst.len, vp(dst_indexes.data), src_indexes.len)
    pipeline.keep(self.data, src_indexes.data, dst.data, dst_indexes.data)
    _put_validity(self, src_indexes, dst, dst_indexes, pipeline)


# Index ranges are checked inside the take / put kernels (out-of-range read → 0, out-of-range write dropped, like the
# reference's robust buffer access) and reported by the next `pipeline.sync()` as ShapeError.  The `_op` forms therefore
# never block; the default forms below synchronise before returning so the error surfaces at the call, as it did when
# the indices were pre-checked with a separate pass.
def _put(self, src_indexes, dst, dst_indexes) -> None:
    p = ArrowComputePipeline(self.get_gpu_device(), "put")
    self.put_op(src_indexes, dst, dst_indexes, p)
    p.finish()
    p.sync()


def _take(self, indexes):
    p = ArrowComputePipeline(self.get_gpu_device(), "take")
    out = self.take_op(indexes, p)
    p.finish()
    p.sync()
    return out


impl(_PRIMS, "merge", _merge_op)
for _c in _PRIMS:
    _c.take_op = _take_op
    _c.take = _take
    _c.put_op = _put_op
    _c.put = _put


# ---- BooleanArrayGPU  [crates/routines/src/bool.rs:48-128]
def _bool_merge_op(self: BooleanArrayGPU, other, mask: BooleanArrayGPU, pipeline: ArrowComputePipeline):
    check_same_len(self, other, "merge")
    check_same_len(self, mask, "merge mask")
    dev = self.gpu_device
    out = dev.create_empty_buffer(max(bitmap_bytes(self.len), 8))
    capi.call("agpu_merge_bits", pipeline._handle, vp(self.data), vp(other.data), vp(mask.data), vp(out), self.len)
    pipeline.keep(self.data, other.data, mask.data, out)
    nulls = merge_null_buffers_op(self.null_buffer, other.null_buffer, mask, pipeline, self.len)
    return BooleanArrayGPU(out, dev, self.len, nulls)


def _bool_take_op(self: BooleanArrayGPU, indexes: UInt32ArrayGPU, pipeline: ArrowComputePipeline):
    dev = self.gpu_device
    out = dev.create_empty_buffer(max(bitmap_bytes(indexes.len), 8))
    capi.call("agpu_take_bits", pipeline._handle, vp(self.data), self.len, vp(indexes.data), vp(out), indexes.len)
    pipeline.keep(self.data, indexes.data, out)
    return BooleanArrayGPU(out, dev, indexes.len, take_null_buffer(self.null_buffer, indexes, pipeline))


def _bool_put_op(self: BooleanArrayGPU, src_indexes, dst: BooleanArrayGPU, dst_indexes, pipeline: ArrowComputePipeline):
    check_same_len(src_indexes, dst_indexes, "put indexes")
    capi.call("agpu_put_bits_bounded", pipeline._handle, vp(self.data), self.len, vp(src_indexes.data), vp(dst.data), dst.len,
              vp(dst_indexes.data), src_indexes.len)
    pipeline.keep(self.data, src_indexes.data, dst.data, dst_indexes.data)
    _put_validity(self, src_indexes, dst, dst_indexes, pipeline)


impl((BooleanArrayGPU,), "merge", _bool_merge_op)
BooleanArrayGPU.take_op = _bool_take_op
BooleanArrayGPU.take = _take
BooleanArrayGPU.put_op = _bool_put_op
BooleanArrayGPU.put = _put

# ---- dyn  [merge.rs:92-140, take.rs:58-94, put.rs:59-110]
_MERGE_DYN = _PRIMS + (BooleanArrayGPU,)
_TAKE_DYN = (Date32ArrayGPU, UInt32ArrayGPU, Int32ArrayGPU, Float32ArrayGPU, BooleanArrayGPU)
_PUT_DYN = (Float32ArrayGPU, Int32ArrayGPU, UInt32ArrayGPU, Date32ArrayGPU, BooleanArrayGPU)


def merge_op_dyn(operand_1: ArrowArrayGPU, operand_2: ArrowArrayGPU, mask: BooleanArrayGPU, pipeline):
    if type(operand_1) is type(operand_2) and type(operand_1) in _MERGE_DYN:
        return operand_1.merge_op(operand_2, mask, pipeline)
    raise OperationNotSupported(
        f"Operation merge_dyn not supported for type {operand_1.get_dtype().name} {operand_2.get_dtype().name}")


def merge_dyn(operand_1, operand_2, mask):
    p = ArrowComputePipeline(operand_1.get_gpu_device(), "merge")
    out = merge_op_dyn(operand_1, operand_2, mask, p)
    p.finish()
    return out


def take_op_dyn(operand_1: ArrowArrayGPU, indexes: UInt32ArrayGPU, pipeline):
    if type(operand_1) in _TAKE_DYN:
        return operand_1.take_op(indexes, pipeline)
    raise OperationNotSupported(f"Operation take_dyn not supported for type {operand_1.get_dtype().name}")


def take_dyn(operand_1, indexes):
    p = ArrowComputePipeline(operand_1.get_gpu_device(), "take")
    out = take_op_dyn(operand_1, indexes, p)
    p.finish()
    p.sync()  # surfaces an out-of-range index here (see _take)
    return out


def take_columns_op(columns, indexes: UInt32ArrayGPU, pipeline: ArrowComputePipeline) -> list:
    """take of the columns of ONE table (primitive arrays of equal length, 1 / 2 / 4-byte values, with or without nulls) by one index
    column — what `[c.take_op(indexes, p) for c in columns]` gives, through agpu_take_columns(_validity): at pipeline sizes everything the
    merge-back take does to the INDEX column runs once for all columns, and a column's validity bit travels with its value (no counterpart
    in the reference, which takes array by array: routines/src/lib.rs:122-143)."""
    import ctypes as C

    if not columns:
        return []
    n_src = columns[0].len
    for c in columns:
        if not isinstance(c, (PrimitiveArrayGpu, BooleanArrayGPU)) or c.len != n_src:
            raise capi.ArrowErrorGPU("ShapeError", "take_columns: primitive or Boolean columns of one length", capi.ERR_SHAPE)
    if any(isinstance(c, BooleanArrayGPU) for c in columns):  # Boolean columns are bitmaps: taken on their own (their pipeline partitions by word)
        prims = [c for c in columns if isinstance(c, PrimitiveArrayGpu)]
        taken = iter(take_columns_op(prims, indexes, pipeline))
        return [c.take_op(indexes, pipeline) if isinstance(c, BooleanArrayGPU) else next(taken) for c in columns]
    dev = columns[0].gpu_device
    k = len(columns)
    outs = [dev.create_empty_buffer(max(indexes.len * c.ITEM_SIZE, 1)) for c in columns]
    widths = (C.c_int32 * k)(*[c.ITEM_SIZE for c in columns])
    vals = (C.c_void_p * k)(*[c.data.ptr for c in columns])
    outp = (C.c_void_p * k)(*[o.ptr for o in outs])
    keep = [indexes.data] + [c.data for c in columns] + outs
    if any(c.null_buffer is not None for c in columns):
        outv = [dev.create_empty_buffer(max(bitmap_bytes(indexes.len), 8)) if c.null_buffer is not None else None for c in columns]
        vbits = (C.c_void_p * k)(*[c.null_buffer.bit_buffer.ptr if c.null_buffer is not None else None for c in columns])
        voutp = (C.c_void_p * k)(*[o.ptr if o is not None else None for o in outv])
        capi.call("agpu_take_columns_validity", pipeline._handle, k, widths, vals, vbits, n_src, vp(indexes.data), outp, voutp, indexes.len)
        keep += [c.null_buffer.bit_buffer for c in columns if c.null_buffer is not None] + [o for o in outv if o is not None]
        nulls = [NullBitBufferGpu(o, indexes.len, dev) if o is not None else None for o in outv]
    else:
        capi.call("agpu_take_columns", pipeline._handle, k, widths, vals, n_src, vp(indexes.data), outp, indexes.len)
        nulls = [None] * k
    pipeline.keep(*keep)
    return [type(c)(o, dev, indexes.len, nb) for c, o, nb in zip(columns, outs, nulls)]


def take_columns(columns, indexes: UInt32ArrayGPU) -> list:
    if not columns:
        return []
    p = ArrowComputePipeline(columns[0].gpu_device)
    out = take_columns_op(columns, indexes, p)
    p.finish()
    return out


def put_op_dyn(src: ArrowArrayGPU, src_indexes, dst: ArrowArrayGPU, dst_indexes, pipeline) -> None:
    if type(src) is type(dst) and type(src) in _PUT_DYN:
        return src.put_op(src_indexes, dst, dst_indexes, pipeline)
    raise OperationNotSupported(
        f"Operation put_dyn not supported for type {src.get_dtype().name} {dst.get_dtype().name}")


def put_dyn(src, src_indexes, dst, dst_indexes) -> None:
    p = ArrowComputePipeline(src.get_gpu_device(), "put")
    put_op_dyn(src, src_indexes, dst, dst_indexes, p)
    p.finish()
    p.sync()


__all__ = ["merge_dyn", "merge_op_dyn", "take_dyn", "take_op_dyn", "put_dyn", "put_op_dyn", "merge_null_buffers_op", "take_columns", "take_columns_op"]
