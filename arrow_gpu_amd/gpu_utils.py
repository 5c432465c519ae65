"""Device runtime: GpuDevice, ArrowComputePipeline, device buffers.

Host-side mirror of `arrow_gpu_array::gpu_utils` (crates/array/src/gpu_utils/{gpu_device,compute_pipeline,
compute_query}.rs) on top of the C ABI.  A GpuDevice is one MI355X; an ArrowComputePipeline is one HIP stream
(ops recorded through `*_op(…, pipeline)` run in order, `finish()` is the submit point and does not wait — same
contract as `queue.submit`, compute_pipeline.rs:259-273); `retrive_data` is the only blocking call
(gpu_device.rs:232-265).
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import threading

import numpy as np

from . import _capi as capi


class DeviceBuffer:
    """Ref-counted HBM allocation (the reference's `Arc<wgpu::Buffer>` / ArrowGpuBuffer, array/buffer.rs:5-53)."""

    __slots__ = ("device", "ptr", "nbytes", "_owned", "__weakref__")

    def __init__(self, device: "GpuDevice", ptr: int, nbytes: int, owned: bool = True):
        self.device = device
        self.ptr = ptr
        self.nbytes = nbytes
        self._owned = owned

    def size(self) -> int:
        return self.nbytes

    def __del__(self):
        if getattr(self, "_owned", False) and self.ptr and self.device is not None and self.device._handle:
            try:
                capi.lib().agpu_free(self.device._handle, C.c_void_p(self.ptr))
            except Exception:
                pass
            self.ptr = 0


class GpuDevice:
    """One gfx950 device.  [ref: GpuDevice::new, gpu_device.rs:46-85]"""

    def __init__(self, ordinal: int = 0):
        h = C.c_void_p()
        capi.call("agpu_device_create", ordinal, C.byref(h))
        self._handle = h
        self.ordinal = ordinal
        self._tls = threading.local()
        name = C.create_string_buffer(128)
        capi.call("agpu_device_name", self._handle, name, 128)
        self.name = name.value.decode()

    @classmethod
    def from_adapter(cls, adapter: int = 0) -> "GpuDevice":
        """`GpuDevice::from_adapter(adapter)` [ref: gpu_device.rs:87-106] named a wgpu adapter; on a ROCm node the choice among
        the visible GPUs is the device ordinal (after HIP_VISIBLE_DEVICES), so the adapter IS the ordinal."""
        return cls(int(adapter))

    # -- default pipeline per host thread (used by immediate calls such as upload / retrive_data)
    def _default_pipeline(self) -> "ArrowComputePipeline":
        p = getattr(self._tls, "pipeline", None)
        if p is None:
            p = ArrowComputePipeline(self, "default")
            self._tls.pipeline = p
        return p

    # -- buffers
    def create_empty_buffer(self, size: int, zero_fill: bool = False, like=()) -> DeviceBuffer:
        """[ref: create_empty_buffer gpu_device.rs:183-192] — wgpu zero-fills; kernels here overwrite every byte.
        `like`: the buffers this one will be read / written together with (an op's inputs, for its output): big blocks are
        then placed against them for the HBM channel hash (agpu_malloc_like)."""
        ptr = C.c_void_p()
        like = [b for b in like if b is not None] if size >= (1 << 30) else ()
        if like:
            arr = (C.c_void_p * len(like))(*[b.ptr for b in like])
            capi.call("agpu_malloc_like", self._handle, size, 1 if zero_fill else 0, arr, len(like), C.byref(ptr))
        else:
            capi.call("agpu_malloc", self._handle, size, 1 if zero_fill else 0, C.byref(ptr))
        return DeviceBuffer(self, ptr.value or 0, size)

    def create_table_buffers(self, sizes, zero_fill: bool = False) -> list:
        """The buffers of one table in ONE device block, placed for the HBM channel hash (`agpu_malloc_table`): columns a
        kernel reads together should not sit in the same hash class.  Each returned DeviceBuffer is an
        ordinary buffer and is freed on its own; the block goes back to the pool with the last of them."""
        n = len(sizes)
        arr = (C.c_uint64 * n)(*[int(x) for x in sizes])
        ptrs = (C.c_void_p * n)()
        capi.call("agpu_malloc_table", self._handle, n, arr, 1 if zero_fill else 0, ptrs)
        return [DeviceBuffer(self, ptrs[k] or 0, int(sizes[k])) for k in range(n)]

    def create_gpu_buffer_with_data(self, data: np.ndarray) -> DeviceBuffer:
        """[ref: create_gpu_buffer_with_data gpu_device.rs:171-181]"""
        data = np.ascontiguousarray(data)
        buf = self.create_empty_buffer(max(data.nbytes, 1))
        if data.nbytes:
            p = self._default_pipeline()
            capi.call("agpu_upload", p._handle, C.c_void_p(buf.ptr), C.c_void_p(data.ctypes.data), data.nbytes)
        return buf

    def create_scalar_buffer(self, value, np_dtype) -> DeviceBuffer:
        """[ref: create_scalar_buffer gpu_device.rs:203-210]"""
        return self.create_gpu_buffer_with_data(np.array([value]).astype(np_dtype))

    def retrive_data(self, buf: DeviceBuffer, nbytes: int | None = None, pipeline=None) -> np.ndarray:
        """Blocking read-back as bytes.  [ref: retrive_data gpu_device.rs:232-265]"""
        nbytes = buf.nbytes if nbytes is None else nbytes
        out = np.empty(nbytes, dtype=np.uint8)
        if nbytes <= capi.MAILBOX_MAX_BYTES:  # (with or without `pipeline`: the device-level wait covers its stream too)
            # a scalar, a small array: the device-level wait delivers it on the way — one wait instead of two
            capi.call("agpu_device_download", self._handle, C.c_void_p(out.ctypes.data), C.c_void_p(buf.ptr if nbytes else 0), nbytes)
            return out
        self.sync()  # other pipelines may still be writing this buffer (the reference polls the whole queue)
        p = pipeline or self._default_pipeline()
        capi.call("agpu_download", p._handle, C.c_void_p(out.ctypes.data), C.c_void_p(buf.ptr), nbytes)
        return out

    def clone_buffer(self, buf: DeviceBuffer, pipeline=None) -> DeviceBuffer:
        """[ref: clone_buffer gpu_device.rs:212-222]"""
        if pipeline is not None:
            return pipeline.clone_buffer(buf)
        # immediate form = record + submit, like the reference's queue.submit inside GpuDevice::clone_buffer: `finish`
        # publishes the copy so pipelines that consume the clone are ordered behind it (and this one behind its producer)
        p = self._default_pipeline()
        out = p.clone_buffer(buf)
        p.finish()
        return out

    def sync(self) -> None:
        capi.call("agpu_device_sync", self._handle)

    def mem_info(self):
        f, t = C.c_uint64(), C.c_uint64()
        capi.call("agpu_device_mem_info", self._handle, C.byref(f), C.byref(t))
        return f.value, t.value

    def __repr__(self):
        return f"GpuDevice(ordinal={self.ordinal}, name={self.name!r})"


class _LazyNode:
    """One recorded element-wise op of a fusing pipeline (see ArrowComputePipeline.record_elementwise)."""

    __slots__ = ("kind", "op", "dtype", "a", "operand", "out", "n", "out_ref")

    def __init__(self, kind, op, dtype, a, operand, out, n, out_ref):
        self.kind, self.op, self.dtype, self.a, self.operand, self.out, self.n, self.out_ref = kind, op, dtype, a, operand, out, n, out_ref


class _ChainStep(C.Structure):
    _fields_ = [("op", C.c_int32), ("kind", C.c_int32), ("operand", C.c_void_p)]


_FUSE_DEFAULT = os.environ.get("AGPU_FUSE", "0") not in ("", "0")
_CHAIN_MAX = 8
_CAST_CHAIN_MAX_ARRAYS = 4  # AGPU_CAST_CHAIN_MAX_ARRAYS (include/arrow_gpu.h)
_UNARY, _SCALAR, _ARRAY = 0, 1, 2
_CMP = 3  # recorded compare (array operand): may only END a chain; `op` is an agpu_cmp_op, `out` the result bitmap
_CAST = 4  # recorded widening cast u8 / i8 / u16 / i16 → f32: may only START a chain; `op` is the SOURCE agpu_dtype, `dtype` F32


class ArrowComputePipeline:
    """Ordered command stream.  [ref: ArrowComputePipeline compute_pipeline.rs:8-22]

    `fuse=True` makes the pipeline behave like the reference's command encoder for same-width
    element-wise ops on f32 / i32 / u32 / Date32 columns: `*_op` calls are only RECORDED and run at `finish()` (or as
    soon as anything else needs the stream), and a run of ops where each consumes the previous result — whose array
    object the caller has already dropped, e.g. `a.add_scalar_op(s, p).mul_scalar_op(s, p)` — is issued as ONE
    `agpu_fused_chain` kernel that never materialises the intermediates.  Intermediates the caller still holds, or that a
    later op reads again, are computed as usual.  As in the reference, results are defined only after `finish()`."""

    def __init__(self, device: GpuDevice, label: str | None = None, hip_stream: int | None = None, fuse: bool | None = None):
        self.device = device
        self.label = label
        h = C.c_void_p()
        if hip_stream is None:
            capi.call("agpu_pipeline_create", device._handle, C.byref(h))
        else:
            capi.call("agpu_pipeline_wrap_stream", device._handle, C.c_void_p(hip_stream), C.byref(h))
        self._h = h
        self._keepalive = []  # buffers referenced by in-flight work (the reference's encoder holds Arc<Buffer>s)
        self.fuse = _FUSE_DEFAULT if fuse is None else bool(fuse)
        self._pending = []    # _LazyNode list (fuse=True only)
        self.stats = {"recorded": 0, "kernels": 0, "fused_chains": 0, "fused_ops": 0}

    # Every launch, copy, sync or timing call reaches the stream through `_handle`: reading it first issues whatever
    # is still only recorded, so eager and recorded work stay in program order.
    @property
    def _handle(self):
        if self._pending:
            self._flush_pending()
        return self._h

    @_handle.setter
    def _handle(self, value):
        self._h = value

    @property
    def _bitmap_handle(self):
        """Raw handle for validity-bitmap work, which never touches the value buffers of recorded ops: no flush."""
        return self._h

    # ---- recording / fusion
    def record_elementwise(self, kind: int, op: int, dtype: int, a: DeviceBuffer, operand, out: DeviceBuffer, n: int) -> _LazyNode:
        node = _LazyNode(kind, op, dtype, a, operand, out, n, None)
        self._pending.append(node)
        self.stats["recorded"] += 1
        return node

    def record_cast(self, from_dtype: int, a: DeviceBuffer, out: DeviceBuffer, n: int) -> _LazyNode:
        """A widening cast to f32 (u8 / i8 / u16 / i16 source): the head of a chain such as `cast → sin`."""
        node = _LazyNode(_CAST, from_dtype, capi.F32, a, None, out, n, None)
        self._pending.append(node)
        self.stats["recorded"] += 1
        return node

    def _launch_single(self, nd: _LazyNode) -> None:
        vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
        if nd.kind == _CMP:
            capi.call("agpu_compare", self._h, nd.op, nd.dtype, vp(nd.a), vp(nd.operand), vp(nd.out), nd.n)
        elif nd.kind == _CAST:
            capi.call("agpu_cast", self._h, nd.op, capi.F32, vp(nd.a), vp(nd.out), nd.n)
        elif nd.kind == _UNARY:
            capi.call("agpu_unary", self._h, nd.op, nd.dtype, vp(nd.a), vp(nd.out), nd.n)
        elif nd.kind == _SCALAR:
            capi.call("agpu_scalar", self._h, nd.op, nd.dtype, vp(nd.a), vp(nd.operand), vp(nd.out), nd.n)
        else:
            capi.call("agpu_binary", self._h, nd.op, nd.dtype, vp(nd.a), vp(nd.operand), vp(nd.out), nd.n)
        self.stats["kernels"] += 1

    def _flush_pending(self) -> None:
        nodes, self._pending = self._pending, []

        def read_later(buf, start):
            return any(m.a is buf or m.operand is buf for m in nodes[start:])

        first_error = None
        i = 0
        while i < len(nodes):
            chain = [nodes[i]]
            while len(chain) < _CHAIN_MAX and chain[-1].kind != _CMP and i + len(chain) < len(nodes):
                last, nxt = chain[-1], nodes[i + len(chain)]
                # dead intermediate: the caller dropped the array object AND nothing else holds its buffer — exactly
                # three references remain (last.out, nxt.a, getrefcount's argument).  A bitcast view sharing the
                # buffer, a later node reading it or a keep-alive entry all add one and keep it materialised.
                dead = (last.out_ref is not None and last.out_ref() is None and sys.getrefcount(last.out) == 3)
                # behind a cast head the kernel reads at most AGPU_CAST_CHAIN_MAX_ARRAYS array operands: the chain is cut
                # in front of the next one (its intermediate is materialised and a plain chain starts there)
                room = not (chain[0].kind == _CAST and nxt.kind == _ARRAY
                            and sum(m.kind == _ARRAY for m in chain) >= _CAST_CHAIN_MAX_ARRAYS)
                if (nxt.a is last.out and nxt.operand is not last.out and nxt.n == last.n and nxt.dtype == last.dtype
                        and nxt.kind != _CAST and not (chain[0].kind == _CAST and nxt.kind == _CMP)  # a cast only starts a chain; cast-headed chains store
                        and room and dead and not read_later(last.out, i + len(chain) + 1)):
                    chain.append(nxt)
                else:
                    break
            # A failing launch must not take the rest of the recording with it: every op was "issued" by the caller, so the
            # remaining chains still run and the first error is raised once the list is empty.
            try:
                self._launch_chain(chain)
            except Exception as e:  # noqa: BLE001 — re-raised below
                if first_error is None:
                    first_error = e
            for nd in chain:
                self.keep(nd.a, nd.operand, nd.out)
            i += len(chain)
        if first_error is not None:
            raise first_error

    def _launch_chain(self, chain) -> None:
        if len(chain) == 1:
            self._launch_single(chain[0])
            return
        try:
            if chain[0].kind == _CAST:  # narrow column in, the f32 chain behind it, one launch (agpu_fused_cast_chain)
                body = chain[1:]
                steps = (_ChainStep * len(body))()
                for k, nd in enumerate(body):
                    steps[k].op, steps[k].kind = nd.op, nd.kind
                    steps[k].operand = nd.operand.ptr if nd.operand is not None else None
                capi.call("agpu_fused_cast_chain", self._h, chain[0].op, C.c_void_p(chain[0].a.ptr), C.cast(steps, C.c_void_p), len(body),
                          C.c_void_p(chain[-1].out.ptr), chain[0].n)
            else:
                body = chain[:-1] if chain[-1].kind == _CMP else chain
                steps = (_ChainStep * len(body))()
                for k, nd in enumerate(body):
                    steps[k].op, steps[k].kind = nd.op, nd.kind
                    steps[k].operand = nd.operand.ptr if nd.operand is not None else None
                if chain[-1].kind == _CMP:  # the chain's value is only compared, never stored
                    cmp = chain[-1]
                    capi.call("agpu_fused_chain_compare", self._h, chain[0].dtype, C.c_void_p(chain[0].a.ptr),
                              C.cast(steps, C.c_void_p), len(body), cmp.op, _ARRAY, C.c_void_p(cmp.operand.ptr),
                              C.c_void_p(cmp.out.ptr), chain[0].n)
                else:
                    capi.call("agpu_fused_chain", self._h, chain[0].dtype, C.c_void_p(chain[0].a.ptr), C.cast(steps, C.c_void_p),
                              len(chain), C.c_void_p(chain[-1].out.ptr), chain[0].n)
        except capi.ArrowErrorGPU as e:
            if getattr(e, "status", None) != capi.ERR_UNSUPPORTED:
                raise
            # a chain shape the fused kernels do not take: the recorded ops one by one (every intermediate has its buffer)
            for nd in chain:
                self._launch_single(nd)
            return
        self.stats["kernels"] += 1
        self.stats["fused_chains"] += 1
        self.stats["fused_ops"] += len(chain)

    def finish(self) -> None:
        """Submit; does NOT wait.  [ref: compute_pipeline.rs:259-273]"""
        capi.call("agpu_pipeline_finish", self._handle)

    def sync(self) -> None:
        capi.call("agpu_pipeline_sync", self._handle)
        self._keepalive.clear()

    def keep(self, *bufs) -> None:
        self._keepalive.extend(b for b in bufs if b is not None)

    def clone_buffer(self, buf: DeviceBuffer, bitmap: bool = False) -> DeviceBuffer:
        """[ref: ArrowComputePipeline::clone_buffer compute_pipeline.rs:275-282]"""
        out = self.device.create_empty_buffer(buf.nbytes)
        self.copy_buffer_to_buffer(buf, 0, out, 0, buf.nbytes, bitmap=bitmap)
        return out

    def copy_buffer_to_buffer(self, src: DeviceBuffer, src_off: int, dst: DeviceBuffer, dst_off: int, size: int,
                              bitmap: bool = False) -> None:
        """[ref: copy_buffer_to_buffer compute_pipeline.rs:284-299]"""
        h = self._bitmap_handle if bitmap else self._handle
        capi.call("agpu_copy", h, C.c_void_p(dst.ptr + dst_off), C.c_void_p(src.ptr + src_off), size)
        self.keep(src, dst)

    def set_tuning(self, key: str, value: int) -> None:
        """Launch tuning of THIS pipeline only (include/arrow_gpu.h: agpu_pipeline_set_tuning)."""
        capi.call("agpu_pipeline_set_tuning", self._h, key.encode(), int(value))

    def wait_pipeline(self, other: "ArrowComputePipeline") -> None:
        """Order this pipeline's later work behind everything `other` has enqueued so far (no host wait)."""
        capi.call("agpu_pipeline_wait_pipeline", self._handle, other._handle)

    def enable_timing(self, bits: int = 2) -> None:
        """[ref: CmpQuery compute_query.rs:7-52] 1 = roctx ranges, 2 = event pair per launch, 4 = wait + log each launch."""
        capi.call("agpu_pipeline_enable_timing", self._h, int(bits))

    def last_kernel_ns(self):
        """(nanoseconds, name) of the last timed launch; blocks on it.  [ref: CmpQuery::wait_for_results :54-75]"""
        ns, name = C.c_uint64(), C.c_char_p()
        capi.call("agpu_pipeline_last_kernel_ns", self._h, C.byref(ns), C.byref(name))
        return ns.value, (name.value or b"").decode()

    def stream(self) -> int:
        s = C.c_void_p()
        capi.call("agpu_pipeline_stream", self._handle, C.byref(s))
        return s.value or 0

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                capi.lib().agpu_pipeline_destroy(h)
            except Exception:
                pass
            self._h = None


class CmpQuery:
    """Pair of HIP events on a pipeline's stream.  [ref: CmpQuery compute_query.rs:7-75 — timestamp query pair]"""

    def __init__(self, device: GpuDevice):
        self.device = device
        self._start, self._stop = C.c_void_p(), C.c_void_p()
        capi.call("agpu_event_create", device._handle, C.byref(self._start))
        capi.call("agpu_event_create", device._handle, C.byref(self._stop))

    def begin(self, pipeline: ArrowComputePipeline) -> None:
        capi.call("agpu_event_record", self._start, pipeline._handle)

    def end(self, pipeline: ArrowComputePipeline) -> None:
        capi.call("agpu_event_record", self._stop, pipeline._handle)

    def wait_for_results(self) -> float:
        """Elapsed milliseconds between begin() and end(); blocks on the stop event."""
        ms = C.c_float()
        capi.call("agpu_event_elapsed_ms", self._start, self._stop, C.byref(ms))
        return ms.value

    def __del__(self):
        for e in (getattr(self, "_start", None), getattr(self, "_stop", None)):
            if e:
                try:
                    capi.lib().agpu_event_destroy(e)
                except Exception:
                    pass


_GPU_DEVICE = None
_GPU_DEVICE_LOCK = threading.Lock()


def gpu_device() -> GpuDevice:
    """Process-wide lazily created device 0 — the reference's `GPU_DEVICE: LazyLock<Arc<GpuDevice>>` (array/src/lib.rs:17).
    Honors LOCAL_RANK when set (one process per GPU)."""
    global _GPU_DEVICE
    with _GPU_DEVICE_LOCK:
        if _GPU_DEVICE is None:
            import os

            _GPU_DEVICE = GpuDevice(int(os.environ.get("LOCAL_RANK", "0")))
        return _GPU_DEVICE
