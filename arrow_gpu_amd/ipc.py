"""Arrow IPC files and streams ↔ GPU arrays, through the C ABI's own reader / writer (`agpu_ipc_*`, csrc/arrow_ipc.hip).

SURVEY §8f-1 names "Arrow C Data Interface / IPC import-export" as the step either side of the hot path; the reference
itself only has `from_slice` / `raw_values` over host Vecs (crates/array/src/array/primitive_array_gpu.rs:22-104).
No pyarrow is needed here: the metadata (Flatbuffers) is parsed and produced by the library, LZ4-compressed bodies
(Feather V2's default) are decompressed by it.  A file is memory-mapped,
so a column travels page cache → HBM without an intermediate host copy; columns of types the GPU has no array for
(utf8, int64, nested …) are skipped and listed with `dtype == -1`.
"""
from __future__ import annotations

import ctypes as C
import mmap
import os

import numpy as np

from . import _capi as capi
from .array import ArrowArrayGPU, NullBitBufferGpu
from .gpu_utils import ArrowComputePipeline, DeviceBuffer, GpuDevice


def _classes():
    from .interop import _class_of_dtype

    return _class_of_dtype()


class IpcField:
    __slots__ = ("name", "format", "dtype", "nullable")

    def __init__(self, name: str, format: str, dtype: int, nullable: bool):  # noqa: A002
        self.name, self.format, self.dtype, self.nullable = name, format, dtype, nullable

    def __repr__(self):
        return f"IpcField({self.name!r}, format={self.format!r}, dtype={self.dtype}, nullable={self.nullable})"


class IpcReader:
    """Reader over a path (memory-mapped), an mmap, bytes or any contiguous buffer.  The source stays referenced (and
    mapped) for the reader's lifetime: column views and uploads read from it directly."""

    def __init__(self, source):
        self._mm = self._file = None
        if isinstance(source, (str, os.PathLike)):
            self._file = open(source, "rb")
            size = os.fstat(self._file.fileno()).st_size
            if size == 0:
                raise capi.ArrowErrorGPU("ShapeError", f"{source}: empty file")
            self._mm = mmap.mmap(self._file.fileno(), 0, access=mmap.ACCESS_READ)
            source = self._mm
        self._view = np.frombuffer(source, dtype=np.uint8)
        self._handle = C.c_void_p()
        capi.call("agpu_ipc_open", C.c_void_p(self._view.ctypes.data), self._view.nbytes, C.byref(self._handle))
        n = C.c_int32()
        capi.call("agpu_ipc_num_fields", self._handle, C.byref(n))
        self.fields = []
        for i in range(n.value):
            f = capi.IpcFieldStruct()
            capi.call("agpu_ipc_field_info", self._handle, i, C.byref(f))
            self.fields.append(IpcField((f.name or b"").decode("utf-8", "replace"), (f.format or b"").decode(), int(f.dtype), bool(f.nullable)))
        nb = C.c_int64()
        capi.call("agpu_ipc_num_batches", self._handle, C.byref(nb))
        self.num_batches = nb.value

    def batch_rows(self, batch: int) -> int:
        r = C.c_int64()
        capi.call("agpu_ipc_batch_rows", self._handle, batch, C.byref(r))
        return r.value

    def column_index(self, name: str) -> int:
        for i, f in enumerate(self.fields):
            if f.name == name:
                return i
        raise KeyError(name)

    def column_view(self, batch: int, column: int):
        """Host view of one column: (values, validity, length, null_count) — numpy views INTO the source bytes (values
        typed by the column's dtype; Boolean data and validity as uint8 bitmaps, LSB first; validity None without nulls).
        Columns of an LZ4-compressed batch are decompressed by the library and come back as copies."""
        c_arr, c_sch = capi.ArrowArrayStruct(), capi.ArrowSchemaStruct()
        capi.call("agpu_ipc_column_view", self._handle, batch, column, C.byref(c_arr), C.byref(c_sch))
        try:
            n, nulls = int(c_arr.length), int(c_arr.null_count)
            dt = self.fields[column].dtype
            np_dt = {capi.F32: np.float32, capi.U32: np.uint32, capi.I32: np.int32, capi.U16: np.uint16, capi.I16: np.int16,
                     capi.U8: np.uint8, capi.I8: np.int8, capi.DATE32: np.int32}.get(dt)
            base = self._view.ctypes.data

            def view(addr, nbytes, dtype):
                off = addr - base
                if 0 <= off and off + nbytes <= self._view.nbytes:
                    return self._view[off:off + nbytes].view(dtype)
                # a compressed (LZ4) batch: the library decompressed into memory the ArrowArray owns — copy before release
                return np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(addr)).copy().view(dtype) if nbytes else np.empty(0, dtype)

            if dt == capi.BOOL:
                values = view(c_arr.buffers[1], (n + 7) // 8, np.uint8)
            else:
                values = view(c_arr.buffers[1], n * np.dtype(np_dt).itemsize, np_dt)
            validity = view(c_arr.buffers[0], (n + 7) // 8, np.uint8) if c_arr.buffers[0] else None
            return values, validity, n, nulls
        finally:
            c_arr.release(C.byref(c_arr))
            c_sch.release(C.byref(c_sch))

    def read_column(self, batch: int, column: int, device: GpuDevice, pipeline: ArrowComputePipeline | None = None) -> ArrowArrayGPU:
        """One column of one record batch → GPU array of the matching type (`agpu_ipc_read_column`)."""
        own = pipeline is None
        p = pipeline or ArrowComputePipeline(device, "ipc.read_column")
        col = capi.ArrowColumnStruct()
        capi.call("agpu_ipc_read_column", self._handle, batch, column, p._handle, C.byref(col))
        n = int(col.length)
        data = DeviceBuffer(device, col.values, int(col.values_bytes))
        nulls = None
        if col.validity:
            nulls = NullBitBufferGpu(DeviceBuffer(device, col.validity, int(col.validity_bytes)), n, device)
        out = _classes()[self.fields[column].dtype](data, device, n, nulls)
        if own:
            p.finish()
            p.sync()
        return out

    def read_batch(self, batch: int, device: GpuDevice, columns=None) -> dict:
        """{name: GPU array} for the readable columns of one record batch (or the named / indexed `columns`).  The batch's
        device buffers come out of ONE block placed for the HBM channel hash (`agpu_ipc_read_batch` →
        `agpu_malloc_table`): columns of a batch are what kernels read together."""
        p = ArrowComputePipeline(device, "ipc.read_batch")
        if columns is None:
            idx = [i for i, f in enumerate(self.fields) if f.dtype >= 0]
        else:
            idx = [c if isinstance(c, int) else self.column_index(c) for c in columns]
        if not idx:
            return {}
        cols = (capi.ArrowColumnStruct * len(idx))()
        capi.call("agpu_ipc_read_batch", self._handle, batch, (C.c_int32 * len(idx))(*idx), len(idx), p._handle, cols)
        out = {}
        for k, i in enumerate(idx):
            col = cols[k]
            n = int(col.length)
            data = DeviceBuffer(device, col.values, int(col.values_bytes))
            nulls = None
            if col.validity:
                nulls = NullBitBufferGpu(DeviceBuffer(device, col.validity, int(col.validity_bytes)), n, device)
            out[self.fields[i].name] = _classes()[self.fields[i].dtype](data, device, n, nulls)
        p.finish()
        p.sync()
        return out

    def read_all(self, device: GpuDevice, columns=None) -> dict:
        """{name: [GPU array per record batch]} — record batches stay separate: they are the sharding / chunking unit."""
        out: dict = {}
        for b in range(self.num_batches):
            for name, arr in self.read_batch(b, device, columns).items():
                out.setdefault(name, []).append(arr)
        return out

    def close(self):
        if getattr(self, "_handle", None):
            capi.lib().agpu_ipc_close(self._handle)
            self._handle = None
        self._view = None
        if self._mm is not None:
            try:
                self._mm.close()
            except BufferError:  # a caller still holds a column view: the mapping goes when the view does
                pass
            self._mm = None
        if self._file is not None:
            self._file.close()
            self._file = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class IpcWriter:
    """Writer of the streaming format (default) or the file format.  `sink`: None → in memory (`finish()` returns bytes),
    a path → that file, an int → an open descriptor.  `fields` = [(name, dtype code or GPU array class, nullable)]."""

    def __init__(self, fields, sink=None, file_format: bool = False, compression: str | None = None):
        if compression not in (None, "lz4"):
            raise capi.OperationNotSupported(f"compression {compression!r}: only 'lz4' (LZ4 frame, Feather V2's default) is implemented")
        codes = {v: k for k, v in _classes().items()}
        self._names = []
        arr = (capi.IpcFieldStruct * max(1, len(fields)))()
        self._keep = []
        for i, (name, dt, nullable) in enumerate(fields):
            code = codes[dt] if isinstance(dt, type) else int(dt)
            b = name.encode()
            self._keep.append(b)
            arr[i].name, arr[i].format, arr[i].dtype, arr[i].nullable = b, None, code, 1 if nullable else 0
            self._names.append(name)
        self._dtypes = [int(arr[i].dtype) for i in range(len(fields))]
        self._own_fd = None
        fd = -1
        if isinstance(sink, (str, os.PathLike)):
            self._own_fd = fd = os.open(sink, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        elif isinstance(sink, int):
            fd = sink
        self._fd = fd
        self._handle = C.c_void_p()
        try:
            capi.call("agpu_ipc_writer_create", arr, len(fields), 1 if file_format else 0, fd, C.byref(self._handle))
            if compression == "lz4":
                capi.call("agpu_ipc_writer_set_compression", self._handle, 1)
        except Exception:
            if self._own_fd is not None:
                os.close(self._own_fd)
                self._own_fd = None
            raise

    def write_batch(self, arrays, pipeline: ArrowComputePipeline | None = None):
        """One record batch from GPU arrays (one per field, schema order): HBM → sink, null counts taken on the GPU."""
        assert len(arrays) == len(self._names)
        cols = (capi.ArrowColumnStruct * max(1, len(arrays)))()
        dev = None
        for i, a in enumerate(arrays):
            dev = a.gpu_device
            cols[i].dtype, cols[i].length, cols[i].null_count = self._dtypes[i], a.len, -1
            cols[i].values, cols[i].values_bytes = a.data.ptr, a.data.nbytes
            if a.null_buffer is not None:
                cols[i].validity, cols[i].validity_bytes = a.null_buffer.bit_buffer.ptr, a.null_buffer.bit_buffer.nbytes
        if dev is None:
            raise capi.ArrowErrorGPU("ArgumentError", "write_batch needs at least one column")
        p = pipeline or ArrowComputePipeline(dev, "ipc.write_batch")
        dev.sync()  # other pipelines may still be writing the arrays (same rule as interop.to_arrow)
        capi.call("agpu_ipc_writer_write_device_batch", self._handle, p._handle, cols)

    def write_host_batch(self, arrays):
        """One record batch from host Arrow arrays (anything with pyarrow's `_export_to_c`): no GPU involved."""
        assert len(arrays) == len(self._names)
        c_arrs = [capi.ArrowArrayStruct() for _ in arrays]
        c_schs = [capi.ArrowSchemaStruct() for _ in arrays]
        ptrs = (C.c_void_p * max(1, len(arrays)))()
        try:
            for i, a in enumerate(arrays):
                a._export_to_c(C.addressof(c_arrs[i]), C.addressof(c_schs[i]))
                ptrs[i] = C.addressof(c_arrs[i])
            capi.call("agpu_ipc_writer_write_batch", self._handle, ptrs)
        finally:
            for x in c_arrs + c_schs:
                if x.release:
                    x.release(C.byref(x))

    def finish(self):
        """Write the end-of-stream marker (and the footer); returns the bytes for an in-memory sink, else None."""
        data, n = C.c_void_p(), C.c_uint64()
        capi.call("agpu_ipc_writer_finish", self._handle, C.byref(data), C.byref(n))
        out = C.string_at(data.value, n.value) if data.value else None
        self.bytes_written = n.value
        self.close()
        return out

    def close(self):
        if getattr(self, "_handle", None):
            capi.lib().agpu_ipc_writer_destroy(self._handle)
            self._handle = None
        if self._own_fd is not None:
            os.close(self._own_fd)
            self._own_fd = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def read_ipc(source, device: GpuDevice, columns=None) -> dict:
    """Path / bytes / mmap → {column name: [GPU array per record batch]}."""
    with IpcReader(source) as r:
        return r.read_all(device, columns)


def write_ipc(columns: dict, sink=None, file_format: bool = False, compression: str | None = None):
    """{name: GPU array | [GPU arrays, one per record batch]} → Arrow IPC (bytes when `sink` is None); compression="lz4"
    writes every buffer as an LZ4 frame (what Feather V2 defaults to)."""
    names = list(columns)
    chunks = [v if isinstance(v, (list, tuple)) else [v] for v in columns.values()]
    nb = len(chunks[0]) if chunks else 0
    assert all(len(c) == nb for c in chunks), "every column needs the same number of record batches"
    fields = [(n, type(c[0]), True) for n, c in zip(names, chunks)]
    w = IpcWriter(fields, sink, file_format, compression)
    for b in range(nb):
        w.write_batch([c[b] for c in chunks])
    return w.finish()
