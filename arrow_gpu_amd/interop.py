"""Arrow interop: ingest / egress of Arrow arrays (pyarrow, or anything speaking the Arrow C Data Interface).

SURVEY §8f-1, the step either side of the hot path: the reference can only build arrays from host `Vec`s
(`from_slice`, `from_optional_slice`: crates/array/src/array/primitive_array_gpu.rs:22-66) and read them back as
`Vec`s (`raw_values`/`values` :68-104).  Here an Arrow array's buffers go to HBM as they are:

  * values buffer  — uploaded from the array's own memory (sliced arrays: from `offset` on), no per-element Python work;
  * validity       — the Arrow bitmap is uploaded once and, when the slice starts at a bit offset, re-aligned ON THE GPU
                     by `agpu_bitmap_copy_bits` (Arrow C Data Interface `offset`);  `null_count == 0` ⇒ no null buffer;
  * staging        — optional page-locked staging (`PinnedStaging`) with asynchronous copies on the pipeline's stream.

`to_arrow` builds a `pyarrow.Array` from downloaded buffers without touching elements; GPU arrays also implement
`__arrow_c_array__`, so `pyarrow.array(gpu_array)` and any other C-Data-Interface consumer work.
pyarrow is imported lazily and only here; the kernels and the C ABI do not depend on it.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi as capi
from .array import (ArrowArrayGPU, BooleanArrayGPU, Date32ArrayGPU, Float32ArrayGPU, Int8ArrayGPU, Int16ArrayGPU,
                    Int32ArrayGPU, NullBitBufferGpu, PrimitiveArrayGpu, UInt8ArrayGPU, UInt16ArrayGPU, UInt32ArrayGPU,
                    bitmap_bytes)
from .gpu_utils import ArrowComputePipeline, DeviceBuffer, GpuDevice


def _pa():
    import pyarrow as pa

    return pa


def _type_map():
    pa = _pa()
    return {
        pa.float32(): Float32ArrayGPU, pa.uint32(): UInt32ArrayGPU, pa.uint16(): UInt16ArrayGPU, pa.uint8(): UInt8ArrayGPU,
        pa.int32(): Int32ArrayGPU, pa.int16(): Int16ArrayGPU, pa.int8(): Int8ArrayGPU, pa.date32(): Date32ArrayGPU,
        pa.bool_(): BooleanArrayGPU,
    }


class PinnedStaging:
    """A page-locked host buffer + async H2D/D2H on a pipeline (hipHostMalloc / hipMemcpyAsync)."""

    def __init__(self, device: GpuDevice, nbytes: int):
        self.device = device
        self.nbytes = nbytes
        ptr = C.c_void_p()
        capi.call("agpu_host_alloc", device._handle, nbytes, C.byref(ptr))
        self.ptr = ptr.value
        self.view = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(self.ptr))

    def upload(self, pipeline: ArrowComputePipeline, dst: DeviceBuffer, nbytes: int, dst_off: int = 0, src_off: int = 0):
        capi.call("agpu_upload_async", pipeline._handle, C.c_void_p(dst.ptr + dst_off), C.c_void_p(self.ptr + src_off), nbytes)

    def download(self, pipeline: ArrowComputePipeline, src: DeviceBuffer, nbytes: int, src_off: int = 0, dst_off: int = 0):
        capi.call("agpu_download_async", pipeline._handle, C.c_void_p(self.ptr + dst_off), C.c_void_p(src.ptr + src_off), nbytes)

    def __del__(self):
        if getattr(self, "ptr", None):
            try:
                capi.lib().agpu_host_free(self.device._handle, C.c_void_p(self.ptr))
            except Exception:
                pass
            self.ptr = None


def _upload_raw(device: GpuDevice, pipeline: ArrowComputePipeline, address: int, nbytes: int, alloc_bytes: int) -> DeviceBuffer:
    buf = device.create_empty_buffer(max(alloc_bytes, 16))
    if nbytes:
        capi.call("agpu_upload", pipeline._handle, C.c_void_p(buf.ptr), C.c_void_p(address), nbytes)
    return buf


def _import_bitmap(device, pipeline, pa_buffer, bit_offset: int, n_bits: int) -> DeviceBuffer:
    """Arrow bitmap (byte-granular, arbitrary bit offset) → word-aligned device bitmap with zero padding."""
    first_byte = bit_offset // 8
    last_byte = (bit_offset + n_bits + 7) // 8
    span = last_byte - first_byte
    # stage the touched bytes at an 8-byte aligned device address, padded to whole words
    staged = device.create_empty_buffer(bitmap_bytes(span * 8) + 8, zero_fill=True)
    if span:
        capi.call("agpu_upload", pipeline._handle, C.c_void_p(staged.ptr), C.c_void_p(pa_buffer.address + first_byte), span)
    out = device.create_empty_buffer(max(bitmap_bytes(n_bits), 8))
    capi.call("agpu_bitmap_copy_bits", pipeline._handle, C.c_void_p(staged.ptr), bit_offset % 8, C.c_void_p(out.ptr), n_bits)
    pipeline.keep(staged, out)
    return out


def from_arrow(obj, device: GpuDevice, pipeline: ArrowComputePipeline | None = None) -> ArrowArrayGPU:
    """pyarrow.Array (or any `__arrow_c_array__` producer) → GPU array of the matching type.  Sliced arrays and
    arrays with or without nulls are handled; a ChunkedArray must go through `from_arrow_chunked`."""
    pa = _pa()
    arr = obj if isinstance(obj, pa.Array) else pa.array(obj)
    cls = _type_map().get(arr.type)
    if cls is None:
        raise capi.OperationNotSupported(f"Arrow type {arr.type} has no GPU array type (f32, u/i 8/16/32, date32, bool)")
    own = pipeline is None
    p = pipeline or ArrowComputePipeline(device, "from_arrow")
    n, off = len(arr), arr.offset
    validity_buf, data_buf = arr.buffers()[0], arr.buffers()[1]
    nulls = None
    if arr.null_count and validity_buf is not None:
        nulls = NullBitBufferGpu(_import_bitmap(device, p, validity_buf, off, n), n, device)
    if cls is BooleanArrayGPU:
        data = _import_bitmap(device, p, data_buf, off, n) if n else device.create_empty_buffer(8)
        out = BooleanArrayGPU(data, device, n, nulls)
    else:
        w = cls.ITEM_SIZE
        data = _upload_raw(device, p, data_buf.address + off * w if n else 0, n * w, n * w)
        out = cls(data, device, n, nulls)
    if own:
        p.sync()
    return out


def from_arrow_chunked(chunked, device: GpuDevice):
    """pyarrow.ChunkedArray → list of GPU arrays, one per chunk (chunks stay separate: that is the sharding unit)."""
    p = ArrowComputePipeline(device, "from_arrow_chunked")
    out = [from_arrow(c, device, p) for c in chunked.chunks]
    p.sync()
    return out


def to_arrow(gpu_array: ArrowArrayGPU):
    """GPU array → pyarrow.Array built from the downloaded buffers (validity bitmap passed through as is)."""
    pa = _pa()
    inv = {v: k for k, v in _type_map().items()}
    typ = inv.get(type(gpu_array))
    if typ is None:
        raise capi.OperationNotSupported(f"{type(gpu_array).__name__} cannot be exported")
    dev = gpu_array.gpu_device
    n = gpu_array.len
    validity = None
    if gpu_array.null_buffer is not None:
        validity = pa.py_buffer(dev.retrive_data(gpu_array.null_buffer.bit_buffer, (n + 7) // 8).tobytes())
    if isinstance(gpu_array, BooleanArrayGPU):
        data = pa.py_buffer(dev.retrive_data(gpu_array.data, (n + 7) // 8).tobytes())
    else:
        data = pa.py_buffer(dev.retrive_data(gpu_array.data, n * gpu_array.ITEM_SIZE).tobytes())
    return pa.Array.from_buffers(typ, n, [validity, data])


def _arrow_c_array(self, requested_schema=None):
    """Arrow PyCapsule interface: lets `pyarrow.array(gpu_array)` / polars / duckdb pull the array."""
    return to_arrow(self).__arrow_c_array__(requested_schema)


for _cls in (PrimitiveArrayGpu, BooleanArrayGPU):
    _cls.__arrow_c_array__ = _arrow_c_array
    _cls.to_arrow = to_arrow
ArrowArrayGPU.from_arrow = staticmethod(from_arrow)
