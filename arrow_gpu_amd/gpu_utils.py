"""Device runtime: GpuDevice, ArrowComputePipeline, device buffers.

Host-side mirror of `arrow_gpu_array::gpu_utils` (crates/array/src/gpu_utils/{gpu_device,compute_pipeline,
compute_query}.rs) on top of the C ABI.  A GpuDevice is one MI355X; an ArrowComputePipeline is one HIP stream
(ops recorded through `*_op(…, pipeline)` run in order, `finish()` is the submit point and does not wait — same
contract as `queue.submit`, compute_pipeline.rs:259-273); `retrive_data` is the only blocking call
(gpu_device.rs:232-265).
"""
from __future__ import annotations

import ctypes as C
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

    # -- default pipeline per host thread (used by immediate calls such as upload / retrive_data)
    def _default_pipeline(self) -> "ArrowComputePipeline":
        p = getattr(self._tls, "pipeline", None)
        if p is None:
            p = ArrowComputePipeline(self, "default")
            self._tls.pipeline = p
        return p

    # -- buffers
    def create_empty_buffer(self, size: int, zero_fill: bool = False) -> DeviceBuffer:
        """[ref: create_empty_buffer gpu_device.rs:183-192] — wgpu zero-fills; kernels here overwrite every byte."""
        ptr = C.c_void_p()
        capi.call("agpu_malloc", self._handle, size, 1 if zero_fill else 0, C.byref(ptr))
        return DeviceBuffer(self, ptr.value or 0, size)

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
        self.sync()  # other pipelines may still be writing this buffer (the reference polls the whole queue)
        p = pipeline or self._default_pipeline()
        capi.call("agpu_download", p._handle, C.c_void_p(out.ctypes.data), C.c_void_p(buf.ptr), nbytes)
        return out

    def clone_buffer(self, buf: DeviceBuffer, pipeline=None) -> DeviceBuffer:
        """[ref: clone_buffer gpu_device.rs:212-222]"""
        p = pipeline or self._default_pipeline()
        return p.clone_buffer(buf)

    def sync(self) -> None:
        capi.call("agpu_device_sync", self._handle)

    def mem_info(self):
        f, t = C.c_uint64(), C.c_uint64()
        capi.call("agpu_device_mem_info", self._handle, C.byref(f), C.byref(t))
        return f.value, t.value

    def __repr__(self):
        return f"GpuDevice(ordinal={self.ordinal}, name={self.name!r})"


class ArrowComputePipeline:
    """Ordered command stream.  [ref: ArrowComputePipeline compute_pipeline.rs:8-22]"""

    def __init__(self, device: GpuDevice, label: str | None = None, hip_stream: int | None = None):
        self.device = device
        self.label = label
        h = C.c_void_p()
        if hip_stream is None:
            capi.call("agpu_pipeline_create", device._handle, C.byref(h))
        else:
            capi.call("agpu_pipeline_wrap_stream", device._handle, C.c_void_p(hip_stream), C.byref(h))
        self._handle = h
        self._keepalive = []  # buffers referenced by in-flight work (the reference's encoder holds Arc<Buffer>s)

    def finish(self) -> None:
        """Submit; does NOT wait.  [ref: compute_pipeline.rs:259-273]"""
        capi.call("agpu_pipeline_finish", self._handle)

    def sync(self) -> None:
        capi.call("agpu_pipeline_sync", self._handle)
        self._keepalive.clear()

    def keep(self, *bufs) -> None:
        self._keepalive.extend(b for b in bufs if b is not None)

    def clone_buffer(self, buf: DeviceBuffer) -> DeviceBuffer:
        """[ref: ArrowComputePipeline::clone_buffer compute_pipeline.rs:275-282]"""
        out = self.device.create_empty_buffer(buf.nbytes)
        self.copy_buffer_to_buffer(buf, 0, out, 0, buf.nbytes)
        return out

    def copy_buffer_to_buffer(self, src: DeviceBuffer, src_off: int, dst: DeviceBuffer, dst_off: int, size: int) -> None:
        """[ref: copy_buffer_to_buffer compute_pipeline.rs:284-299]"""
        capi.call("agpu_copy", self._handle, C.c_void_p(dst.ptr + dst_off), C.c_void_p(src.ptr + src_off), size)
        self.keep(src, dst)

    def stream(self) -> int:
        s = C.c_void_p()
        capi.call("agpu_pipeline_stream", self._handle, C.byref(s))
        return s.value or 0

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            try:
                capi.lib().agpu_pipeline_destroy(h)
            except Exception:
                pass
            self._handle = None


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
