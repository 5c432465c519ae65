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


def _release(c_struct):
    if c_struct.release:
        c_struct.release(C.byref(c_struct))


def _class_of_dtype():
    return {capi.F32: Float32ArrayGPU, capi.U32: UInt32ArrayGPU, capi.U16: UInt16ArrayGPU, capi.U8: UInt8ArrayGPU,
            capi.I32: Int32ArrayGPU, capi.I16: Int16ArrayGPU, capi.I8: Int8ArrayGPU, capi.DATE32: Date32ArrayGPU,
            capi.BOOL: BooleanArrayGPU}


def from_arrow(obj, device: GpuDevice, pipeline: ArrowComputePipeline | None = None) -> ArrowArrayGPU:
    """pyarrow.Array (or any `__arrow_c_array__` producer) → GPU array of the matching type, through the C ABI's
    Arrow C Data Interface entry point: the producer exports its `ArrowArray` / `ArrowSchema` pair, `agpu_import_arrow`
    consumes the buffers as they are — values from `offset` on, bitmaps re-aligned on the GPU, arrays ≥ 1 MiB through
    the threaded page-locked staging — and the pair is released.  A ChunkedArray goes through `from_arrow_chunked`."""
    pa = _pa()
    arr = obj if isinstance(obj, pa.Array) else pa.array(obj)
    if arr.type not in _type_map():
        raise capi.OperationNotSupported(f"Arrow type {arr.type} has no GPU array type (f32, u/i 8/16/32, date32, bool)")
    own = pipeline is None
    p = pipeline or ArrowComputePipeline(device, "from_arrow")
    c_arr, c_sch, col = capi.ArrowArrayStruct(), capi.ArrowSchemaStruct(), capi.ArrowColumnStruct()
    arr._export_to_c(C.addressof(c_arr), C.addressof(c_sch))
    try:
        capi.call("agpu_import_arrow", p._handle, C.byref(c_arr), C.byref(c_sch), C.byref(col))
    finally:
        _release(c_arr)
        _release(c_sch)
    n = int(col.length)
    data = DeviceBuffer(device, col.values, int(col.values_bytes))
    nulls = None
    if col.validity:
        nulls = NullBitBufferGpu(DeviceBuffer(device, col.validity, int(col.validity_bytes)), n, device)
    cls = _type_map()[arr.type]  # date32 and i32 share storage: the Arrow type decides
    out = cls(data, device, n, nulls)
    if own:
        p.finish()
        p.sync()
    return out


def from_arrow_batch(batch, device: GpuDevice, pipeline: ArrowComputePipeline | None = None) -> dict:
    """pyarrow.RecordBatch (or {name: pyarrow.Array}) → {name: GPU array}, the device buffers of all columns out of ONE block
    placed for the HBM channel hash (`agpu_import_arrow_table`): columns of a batch are what kernels read together."""
    pa = _pa()
    items = list(zip(batch.schema.names, batch.columns)) if isinstance(batch, pa.RecordBatch) else list(batch.items())
    for name, arr in items:
        if arr.type not in _type_map():
            raise capi.OperationNotSupported(f"column {name!r}: Arrow type {arr.type} has no GPU array type")
    n = len(items)
    own = pipeline is None
    p = pipeline or ArrowComputePipeline(device, "from_arrow_batch")
    c_arrs, c_schs = [capi.ArrowArrayStruct() for _ in items], [capi.ArrowSchemaStruct() for _ in items]
    ap, sp = (C.c_void_p * n)(), (C.c_void_p * n)()
    cols = (capi.ArrowColumnStruct * n)()
    try:
        for k, (_, arr) in enumerate(items):
            arr._export_to_c(C.addressof(c_arrs[k]), C.addressof(c_schs[k]))
            ap[k], sp[k] = C.addressof(c_arrs[k]), C.addressof(c_schs[k])
        capi.call("agpu_import_arrow_table", p._handle, n, ap, sp, cols)
    finally:
        for x in c_arrs + c_schs:
            _release(x)
    out = {}
    for k, (name, arr) in enumerate(items):
        col = cols[k]
        ln = int(col.length)
        data = DeviceBuffer(device, col.values, int(col.values_bytes))
        nulls = NullBitBufferGpu(DeviceBuffer(device, col.validity, int(col.validity_bytes)), ln, device) if col.validity else None
        out[name] = _type_map()[arr.type](data, device, ln, nulls)
    if own:
        p.finish()
        p.sync()
    return out


def from_arrow_reader(reader, device: GpuDevice, columns=None):
    """pyarrow.RecordBatchReader (or anything with `_export_to_c` for the Arrow C STREAM interface) → generator of
    {name: GPU array} per record batch, pulled through `agpu_import_arrow_stream_next`: the stream is consumed by the
    library batch by batch, each batch's buffers in one table-placed device block."""
    names = list(reader.schema.names)
    tm = _type_map()
    if columns is None:
        idx = [i for i, f in enumerate(reader.schema) if f.type in tm]
    else:
        idx = [c if isinstance(c, int) else names.index(c) for c in columns]
    types = [reader.schema.field(i).type for i in idx]
    stream = capi.ArrowArrayStreamStruct()
    reader._export_to_c(C.addressof(stream))
    p = ArrowComputePipeline(device, "from_arrow_reader")
    try:
        while True:
            cols = (capi.ArrowColumnStruct * len(idx))()
            rows = C.c_int64()
            capi.call("agpu_import_arrow_stream_next", p._handle, C.byref(stream), (C.c_int32 * len(idx))(*idx), len(idx), cols, C.byref(rows))
            if rows.value < 0:
                break
            out = {}
            for k, i in enumerate(idx):
                col = cols[k]
                ln = int(col.length)
                data = DeviceBuffer(device, col.values, int(col.values_bytes))
                nulls = NullBitBufferGpu(DeviceBuffer(device, col.validity, int(col.validity_bytes)), ln, device) if col.validity else None
                out[names[i]] = tm[types[k]](data, device, ln, nulls)
            p.finish()
            p.sync()
            yield out
    finally:
        if stream.release:
            stream.release(C.addressof(stream))


def from_arrow_chunked(chunked, device: GpuDevice):
    """pyarrow.ChunkedArray → list of GPU arrays, one per chunk (chunks stay separate: that is the sharding unit)."""
    p = ArrowComputePipeline(device, "from_arrow_chunked")
    out = [from_arrow(c, device, p) for c in chunked.chunks]
    p.sync()
    return out


def to_arrow(gpu_array: ArrowArrayGPU, pipeline: ArrowComputePipeline | None = None):
    """GPU array → pyarrow.Array: `agpu_export_arrow` downloads values and validity into freshly allocated host buffers
    behind an `ArrowArray` / `ArrowSchema` pair whose release callbacks free them; pyarrow imports (and later releases)
    the pair — no element is touched in Python."""
    pa = _pa()
    dev = gpu_array.gpu_device
    codes = {v: k for k, v in _class_of_dtype().items()}
    if type(gpu_array) not in codes:
        raise capi.OperationNotSupported(f"{type(gpu_array).__name__} cannot be exported")
    col = capi.ArrowColumnStruct()
    col.dtype, col.length, col.null_count = codes[type(gpu_array)], gpu_array.len, -1
    col.values, col.values_bytes = gpu_array.data.ptr, gpu_array.data.nbytes
    if gpu_array.null_buffer is not None:
        col.validity, col.validity_bytes = gpu_array.null_buffer.bit_buffer.ptr, gpu_array.null_buffer.bit_buffer.nbytes
    else:
        col.null_count = 0
    p = pipeline or ArrowComputePipeline(dev, "to_arrow")
    dev.sync()  # other pipelines may still be writing the array (the reference's read-back polls the whole queue)
    if gpu_array.null_buffer is not None and gpu_array.null_buffer.null_count_known():
        col.null_count = gpu_array.null_buffer.null_count()  # left behind by the kernel that made the bitmap: an 8-byte read, no pass
    c_arr, c_sch = capi.ArrowArrayStruct(), capi.ArrowSchemaStruct()
    capi.call("agpu_export_arrow", p._handle, C.byref(col), C.byref(c_arr), C.byref(c_sch))
    return pa.Array._import_from_c(C.addressof(c_arr), C.addressof(c_sch))


def _arrow_c_array(self, requested_schema=None):
    """Arrow PyCapsule interface: lets `pyarrow.array(gpu_array)` / polars / duckdb pull the array."""
    return to_arrow(self).__arrow_c_array__(requested_schema)


for _cls in (PrimitiveArrayGpu, BooleanArrayGPU):
    _cls.__arrow_c_array__ = _arrow_c_array
    _cls.to_arrow = to_arrow
ArrowArrayGPU.from_arrow = staticmethod(from_arrow)


def map_chunks(device: GpuDevice, inputs, out: np.ndarray, chunk_rows: int, launch) -> dict:
    """Host → HBM → host with everything overlapped (SURVEY §8f-1 "async H2D overlap"): the columns in `inputs` (numpy
    arrays of equal length, pageable host memory) are cut into chunks of `chunk_rows`; an uploader thread stages chunk
    k+1 into one of two device buffer sets while the compute pipeline runs `launch(pipeline, device_inputs, device_out,
    rows)` on chunk k and streams its result back into `out`.  The upload and the compute + download run on different
    pipelines (HIP streams) tied together by `wait_pipeline`; H2D and D2H use opposite directions of the link.  The
    reference has no counterpart: it uploads whole Vecs before the first dispatch and reads back after the last
    (primitive_array_gpu.rs:22-74).  Returns timing facts."""
    import threading
    import time

    n = len(out)
    assert all(len(a) == n for a in inputs)
    up, comp = ArrowComputePipeline(device, "map_chunks.upload"), ArrowComputePipeline(device, "map_chunks.compute")
    sets = []
    for _ in range(2):
        ins = [device.create_empty_buffer(chunk_rows * a.dtype.itemsize) for a in inputs]
        sets.append((ins, device.create_empty_buffer(chunk_rows * out.dtype.itemsize)))
    free = [threading.Semaphore(1), threading.Semaphore(1)]
    ready = [threading.Semaphore(0), threading.Semaphore(0)]
    nchunks = (n + chunk_rows - 1) // chunk_rows
    errors = []

    stop = threading.Event()

    def uploader():
        try:
            for k in range(nchunks):
                s = k % 2
                free[s].acquire()
                if stop.is_set():  # the consumer failed: it released both `free` slots to get us out of the acquire above
                    return
                r0, rows = k * chunk_rows, min(chunk_rows, n - k * chunk_rows)
                for a, buf in zip(inputs, sets[s][0]):
                    src = a[r0:r0 + rows]
                    capi.call("agpu_staged_copy", up._handle, C.c_void_p(buf.ptr), C.c_void_p(src.ctypes.data), src.nbytes, 1)
                ready[s].release()
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            for s in ready:
                s.release()

    t0 = time.perf_counter()
    th = threading.Thread(target=uploader, daemon=True)
    th.start()
    failed = True
    try:
        for k in range(nchunks):
            s = k % 2
            ready[s].acquire()
            if errors:
                break
            r0, rows = k * chunk_rows, min(chunk_rows, n - k * chunk_rows)
            comp.wait_pipeline(up)  # the chunk's last DMA may still be in flight on the upload stream
            launch(comp, sets[s][0], sets[s][1], rows)
            dst = out[r0:r0 + rows]
            capi.call("agpu_staged_copy", comp._handle, C.c_void_p(sets[s][1].ptr), C.c_void_p(dst.ctypes.data), dst.nbytes, 0)
            free[s].release()  # the download has drained: the set may be refilled
        failed = False
    finally:
        # whatever happened above (launch() raised, a copy failed, the uploader reported an error): the uploader must not
        # stay parked in free[s].acquire() — a non-daemon thread blocked there kept the interpreter from exiting
        stop.set()
        for sem in free:
            sem.release()
        th.join()
        if failed:  # the device buffer sets are about to be dropped: nothing may still be reading them
            try:
                comp.sync()
                up.sync()
            except Exception:  # noqa: BLE001 — the original exception is the one to report
                pass
        else:
            comp.sync()
    dt = time.perf_counter() - t0
    if errors:
        raise errors[0]
    moved = sum(a.nbytes for a in inputs) + out.nbytes
    return {"seconds": dt, "chunks": nchunks, "host_bytes_moved": moved, "GBps_host_bytes": moved / dt / 1e9}
