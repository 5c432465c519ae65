"""arrow_gpu_amd — MI355X-native Arrow columnar compute kernels behind psvri/arrow-gpu's API surface.

`import arrow_gpu_amd as ag` gives the flat namespace of the reference's umbrella crate `arrow_gpu`
(crates/arrow/src/lib.rs:1-3): the array types (`ag.Float32ArrayGPU`, `ag.BooleanArrayGPU`, …), the device runtime
(`ag.GpuDevice`, `ag.ArrowComputePipeline`, `ag.GPU_DEVICE()`), and `ag.kernels.*` / the `*_dyn` free functions.
Every op runs as a hand-written HIP kernel for gfx950 through the C ABI in include/arrow_gpu.h
(arrow_gpu_amd/lib/libarrow_gpu_hip.so).  There is no CPU fallback.
"""
from . import kernels  # noqa: F401  (attaches the op methods to the array classes)
from ._capi import ArrowErrorGPU, CastingNotSupported, OperationNotSupported  # noqa: F401
from .array import (ARRAY_OF_TYPE, ArrowArrayGPU, ArrowType, BooleanArrayGPU, BooleanBufferBuilder,  # noqa: F401
                    Date32ArrayGPU, Float32ArrayGPU, Int8ArrayGPU, Int16ArrayGPU, Int32ArrayGPU, NullBitBufferGpu,
                    Operand, PrimitiveArrayGpu, ScalarArray, ScalarValue, UInt8ArrayGPU, UInt16ArrayGPU, UInt32ArrayGPU,
                    broadcast_dyn, broadcast_op_dyn)
from .gpu_utils import ArrowComputePipeline, CmpQuery, DeviceBuffer, GpuDevice  # noqa: F401
from .gpu_utils import gpu_device as GPU_DEVICE  # noqa: F401
from .kernels import *  # noqa: F401,F403
from . import interop, ipc, sharding  # noqa: F401,E402
from .interop import PinnedStaging, from_arrow, from_arrow_batch, from_arrow_chunked, from_arrow_reader, to_arrow  # noqa: F401,E402
