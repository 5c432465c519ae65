"""Array model: PrimitiveArrayGpu<T>, BooleanArrayGPU, NullBitBufferGpu, BooleanBufferBuilder, ArrowType.

Host-side mirror of `arrow_gpu_array::array` (crates/array/src/array/{mod,primitive_array_gpu,boolean_gpu,
null_bit_buffer,*_gpu}.rs).  Same names, same construction rules (null slots hold default()=0, validity is an
LSB-first bitmap, `from_optional_slice` always materialises a validity bitmap), same read-back API
(`raw_values`, `values`).  Rust's `ArrowArrayGPU` enum becomes the common base class: `x.into()` is the identity and
`T::try_from(dyn_array)` is `T.try_from(dyn_array)`.
"""
from __future__ import annotations

import ctypes as C
import enum
from typing import Optional, Sequence

import numpy as np

from . import _capi as capi
from ._capi import ArrowErrorGPU, CastingNotSupported
from .gpu_utils import ArrowComputePipeline, DeviceBuffer, GpuDevice


class ArrowType(enum.Enum):
    """[ref: crates/array/src/array/mod.rs:40-50]"""

    BooleanType = capi.BOOL
    Float32Type = capi.F32
    UInt32Type = capi.U32
    UInt16Type = capi.U16
    UInt8Type = capi.U8
    Int32Type = capi.I32
    Int16Type = capi.I16
    Int8Type = capi.I8
    Date32Type = capi.DATE32


def bitmap_bytes(n_bits: int) -> int:
    """Allocation size of a bitmap handed to kernels: whole 64-bit words (include/arrow_gpu.h conventions)."""
    return (n_bits + 63) // 64 * 8


class BooleanBufferBuilder:
    """Host-side bitmap builder.  [ref: crates/array/src/array/null_bit_buffer.rs:10-62]"""

    def __init__(self, size: int = 1024, _set: bool = False):
        aligned = (size + 7) // 8
        self.len = size
        if _set:
            self.data = np.full(aligned, 0xFF, dtype=np.uint8)
            diff = size % 8
            if diff != 0:
                self.data[aligned - 1] = 0xFF >> (8 - diff)  # padding bits zero
            self.contains_nulls = False
        else:
            self.data = np.zeros(aligned, dtype=np.uint8)
            self.contains_nulls = True

    @classmethod
    def new(cls):
        return cls(1024)

    @classmethod
    def new_with_capacity(cls, size: int):
        return cls(size)

    @classmethod
    def new_set_with_capacity(cls, size: int):
        return cls(size, _set=True)

    def set_bit(self, pos: int) -> None:
        self.data[pos // 8] |= np.uint8(1 << (pos % 8))

    def unset_bit(self, pos: int) -> None:
        self.data[pos // 8] &= np.uint8(~(1 << (pos % 8)) & 0xFF)

    def is_set(self, pos: int) -> bool:
        return bool(self.data[pos // 8] & (1 << (pos % 8)))

    @staticmethod
    def is_set_in_slice(data, pos: int) -> bool:
        return bool(data[pos // 8] & (1 << (pos % 8)))

    @classmethod
    def from_bools(cls, bools) -> "BooleanBufferBuilder":
        b = np.asarray(bools, dtype=bool)
        out = cls(len(b))
        packed = np.packbits(b, bitorder="little")
        out.data[: len(packed)] = packed
        return out


def _upload_bitmap(device: GpuDevice, data: np.ndarray, n_bits: int) -> DeviceBuffer:
    """Upload a byte-granular bitmap into a whole-word device buffer (padding bytes zero)."""
    padded = np.zeros(max(bitmap_bytes(n_bits), 8), dtype=np.uint8)
    padded[: len(data)] = data
    return device.create_gpu_buffer_with_data(padded)


class NullBitBufferGpu:
    """Validity bitmap in HBM.  [ref: crates/array/src/array/null_bit_buffer.rs:92-243]"""

    def __init__(self, bit_buffer: DeviceBuffer, len_: int, gpu_device: GpuDevice, count_buf: Optional[DeviceBuffer] = None,
                 count_is_set_bits: bool = False):
        self.bit_buffer = bit_buffer
        self.len = len_
        self.gpu_device = gpu_device
        # the kernel that produced this bitmap may have left its count behind as a by-product (a device u64: the null count,
        # or the number of set bits): `null_count()` then costs one 8-byte read-back instead of a pass over the bitmap
        self._count_buf = count_buf
        self._count_is_set_bits = count_is_set_bits
        self._null_count: Optional[int] = None

    def null_count_known(self) -> bool:
        """True when `null_count()` needs no pass over the bitmap (already read, or left behind by the producing kernel)."""
        return self._null_count is not None or self._count_buf is not None

    def null_count(self) -> int:
        """Number of null slots (unset bits among the first `len`).  Blocking.  The reference has no such accessor: its
        only bit count is countob + Sum over a Boolean array [ref: crates/logical/src/boolean.rs:120-146]."""
        if self._null_count is None:
            if self._count_buf is None:
                self._count_buf = self.gpu_device.create_empty_buffer(8)
                self._count_is_set_bits = True
                p = self.gpu_device._default_pipeline()
                capi.call("agpu_bitmap_popcount", p._handle, C.c_void_p(self.bit_buffer.ptr), self.len, C.c_void_p(self._count_buf.ptr))
            v = int(self.gpu_device.retrive_data(self._count_buf, 8).view(np.uint64)[0])
            self._null_count = self.len - v if self._count_is_set_bits else v
        return self._null_count

    @classmethod
    def new(cls, gpu_device: GpuDevice, builder: BooleanBufferBuilder) -> Optional["NullBitBufferGpu"]:
        if builder.contains_nulls:
            return cls(_upload_bitmap(gpu_device, builder.data, builder.len), builder.len, gpu_device)
        return None

    @classmethod
    def new_set_with_capacity(cls, gpu_device: GpuDevice, size: int) -> "NullBitBufferGpu":
        b = BooleanBufferBuilder.new_set_with_capacity(size)
        return cls(_upload_bitmap(gpu_device, b.data, size), size, gpu_device)

    def raw_values(self) -> np.ndarray:
        raw = self.gpu_device.retrive_data(self.bit_buffer)
        return raw[: (self.len + 7) // 8].copy()

    # -- clone
    @staticmethod
    def clone_null_bit_buffer(data: Optional["NullBitBufferGpu"]) -> Optional["NullBitBufferGpu"]:
        if data is None:
            return None
        return NullBitBufferGpu(data.gpu_device.clone_buffer(data.bit_buffer), data.len, data.gpu_device)

    @staticmethod
    def clone_null_bit_buffer_op(data, pipeline: ArrowComputePipeline):
        if data is None:
            return None
        return NullBitBufferGpu(pipeline.clone_buffer(data.bit_buffer, bitmap=True), data.len, data.gpu_device)

    clone_null_bit_buffer_pass = clone_null_bit_buffer_op

    # -- merge (the validity AND)
    @staticmethod
    def merge_null_bit_buffer_op(left, right, pipeline: ArrowComputePipeline):
        """(None,None)→None; one side → copy; both → AND.  [ref: null_bit_buffer.rs:206-243]"""
        if left is None and right is None:
            return None
        if left is None or right is None:
            x = left if left is not None else right
            return NullBitBufferGpu(pipeline.clone_buffer(x.bit_buffer, bitmap=True), x.len, x.gpu_device)
        assert left.len == right.len, "validity bitmaps of different length"
        assert left.gpu_device is right.gpu_device
        out = left.gpu_device.create_empty_buffer(left.bit_buffer.nbytes)
        cnt = left.gpu_device.create_empty_buffer(8)  # set bits of the result, counted by the waves that store it
        capi.call("agpu_bitmap_binary_count", pipeline._bitmap_handle, capi.OP_AND, C.c_void_p(left.bit_buffer.ptr),
                  C.c_void_p(right.bit_buffer.ptr), C.c_void_p(out.ptr), left.len, C.c_void_p(cnt.ptr))
        pipeline.keep(left.bit_buffer, right.bit_buffer, out, cnt)
        return NullBitBufferGpu(out, left.len, left.gpu_device, cnt, True)

    @staticmethod
    def merge_null_bit_buffer(left, right):
        dev = (left or right).gpu_device if (left or right) else None
        if dev is None:
            return None
        p = ArrowComputePipeline(dev, "merge_null_bit_buffer")
        out = NullBitBufferGpu.merge_null_bit_buffer_op(left, right, p)
        p.finish()
        return out


class ArrowArrayGPU:
    """Common base = Rust's `enum ArrowArrayGPU` (crates/array/src/array/mod.rs:104-186)."""

    ARROW_TYPE: ArrowType
    DTYPE: int

    def get_gpu_device(self) -> GpuDevice:
        return self.gpu_device

    def get_dtype(self) -> ArrowType:
        return self.ARROW_TYPE

    def get_raw_values(self) -> "ScalarArray":
        """`ArrowArrayGPU::get_raw_values` → `ScalarArray` [ref: crates/array/src/array/mod.rs:145-157]: the raw values tagged with their
        element type (`ScalarArray.F32Vec(…)`, `ScalarArray.BOOLVec(…)`); compares equal to a plain list / ndarray of the same values."""
        return ScalarArray.of(type(self), self.raw_values())

    def into(self) -> "ArrowArrayGPU":
        return self

    @classmethod
    def try_from(cls, value: "ArrowArrayGPU"):
        """`TryFrom<ArrowArrayGPU>`: CastingNotSupported on a variant mismatch (e.g. f32_gpu.rs:45-57)."""
        if type(value) is cls:
            return value
        raise CastingNotSupported(f"could not cast {type(value).__name__} into {cls.__name__}")

    def __len__(self) -> int:
        return self.len


class PrimitiveArrayGpu(ArrowArrayGPU):
    """Arrow primitive array in HBM.  [ref: crates/array/src/array/primitive_array_gpu.rs:12-117]"""

    NP_DTYPE = None
    ITEM_SIZE = 0

    def __init__(self, data: DeviceBuffer, gpu_device: GpuDevice, len_: int, null_buffer: Optional[NullBitBufferGpu]):
        self.data = data
        self.gpu_device = gpu_device
        self.len = len_
        self.null_buffer = null_buffer

    # -- construction
    @classmethod
    def from_optional_slice(cls, value: Sequence, gpu_device: GpuDevice):
        n = len(value)
        valid = np.fromiter((v is not None for v in value), dtype=bool, count=n)
        host = np.array([cls._to_native(v) if v is not None else 0 for v in value], dtype=cls.NP_DTYPE).reshape(n)
        builder = BooleanBufferBuilder.new_with_capacity(n)
        packed = np.packbits(valid, bitorder="little")  # bit i at byte i/8, mask 1 << (i % 8); null slots hold 0
        builder.data[: len(packed)] = packed
        data = gpu_device.create_gpu_buffer_with_data(host)
        return cls(data, gpu_device, n, NullBitBufferGpu.new(gpu_device, builder))

    @classmethod
    def from_slice(cls, value, gpu_device: GpuDevice):
        if isinstance(value, np.ndarray) and value.dtype == cls.NP_DTYPE:
            host = np.ascontiguousarray(value)
        else:
            host = np.array([cls._to_native(v) for v in value], dtype=cls.NP_DTYPE)
        return cls(gpu_device.create_gpu_buffer_with_data(host), gpu_device, len(host), None)

    @classmethod
    def _to_native(cls, v):
        if np.issubdtype(cls.NP_DTYPE, np.integer):
            info = np.iinfo(cls.NP_DTYPE)
            v = int(v)
            if not (info.min <= v <= info.max):  # wrap like a Rust `as` cast
                v = (v - info.min) % (1 << info.bits) + info.min
            return cls.NP_DTYPE(v)
        return cls.NP_DTYPE(v)

    # -- read back
    def raw_values(self) -> np.ndarray:
        raw = self.gpu_device.retrive_data(self.data, self.len * self.ITEM_SIZE)
        return raw.view(self.NP_DTYPE)[: self.len].copy()

    def values(self) -> list:
        vals = self.raw_values()
        if self.null_buffer is None:
            return vals.tolist()
        valid = np.unpackbits(self.null_buffer.raw_values(), bitorder="little")[: self.len].tolist()
        return [v if ok else None for v, ok in zip(vals.tolist(), valid)]

    def clone_array(self):
        data = self.gpu_device.clone_buffer(self.data)
        return type(self)(data, self.gpu_device, self.len, NullBitBufferGpu.clone_null_bit_buffer(self.null_buffer))

    # -- Broadcast<T>  [ref: crates/array/src/kernels/broadcast.rs:6-17, array/src/array/f32_gpu.rs:14-37]
    @classmethod
    def broadcast_op(cls, value, len_: int, pipeline: ArrowComputePipeline):
        dev = pipeline.device
        out = dev.create_empty_buffer(max(len_ * cls.ITEM_SIZE, 1))
        raw = np.zeros(4, dtype=np.uint8)
        vb = np.array([cls._to_native(value)], dtype=cls.NP_DTYPE).view(np.uint8)
        raw[: len(vb)] = vb
        capi.call("agpu_broadcast", pipeline._handle, cls.DTYPE, int(raw.view(np.uint32)[0]), C.c_void_p(out.ptr), len_)
        pipeline.keep(out)
        return cls(out, dev, len_, None)

    @classmethod
    def broadcast(cls, value, len_: int, gpu_device: GpuDevice):
        p = ArrowComputePipeline(gpu_device, "broadcast")
        arr = cls.broadcast_op(value, len_, p)
        p.finish()
        return arr

    def __repr__(self):
        return f"{type(self).__name__}(len={self.len}, nulls={'yes' if self.null_buffer else 'no'})"


def _primitive(name: str, arrow_type: ArrowType, np_dtype, item_size: int):
    return type(name, (PrimitiveArrayGpu,), {
        "ARROW_TYPE": arrow_type, "DTYPE": arrow_type.value, "NP_DTYPE": np_dtype, "ITEM_SIZE": item_size,
        "__doc__": f"{name} = PrimitiveArrayGpu<{np.dtype(np_dtype).name}> (crates/array/src/array/*_gpu.rs)",
    })


Float32ArrayGPU = _primitive("Float32ArrayGPU", ArrowType.Float32Type, np.float32, 4)
UInt32ArrayGPU = _primitive("UInt32ArrayGPU", ArrowType.UInt32Type, np.uint32, 4)
UInt16ArrayGPU = _primitive("UInt16ArrayGPU", ArrowType.UInt16Type, np.uint16, 2)
UInt8ArrayGPU = _primitive("UInt8ArrayGPU", ArrowType.UInt8Type, np.uint8, 1)
Int32ArrayGPU = _primitive("Int32ArrayGPU", ArrowType.Int32Type, np.int32, 4)
# NB the reference declares ITEM_SIZE = 4 for i16 (array/src/array/mod.rs:83) — a bug not reproduced here.
Int16ArrayGPU = _primitive("Int16ArrayGPU", ArrowType.Int16Type, np.int16, 2)
Int8ArrayGPU = _primitive("Int8ArrayGPU", ArrowType.Int8Type, np.int8, 1)
Date32ArrayGPU = _primitive("Date32ArrayGPU", ArrowType.Date32Type, np.int32, 4)


class BooleanArrayGPU(ArrowArrayGPU):
    """Bit-packed Boolean array.  [ref: crates/array/src/array/boolean_gpu.rs:15-135]"""

    ARROW_TYPE = ArrowType.BooleanType
    DTYPE = capi.BOOL

    def __init__(self, data: DeviceBuffer, gpu_device: GpuDevice, len_: int, null_buffer: Optional[NullBitBufferGpu]):
        self.data = data
        self.gpu_device = gpu_device
        self.len = len_
        self.null_buffer = null_buffer

    @classmethod
    def from_optional_slice(cls, value: Sequence, gpu_device: GpuDevice):
        n = len(value)
        buf = BooleanBufferBuilder.new_with_capacity(n)
        nulls = BooleanBufferBuilder.new_with_capacity(n)
        for i, v in enumerate(value):
            if v is True:
                buf.set_bit(i)
                nulls.set_bit(i)
            elif v is False:
                nulls.set_bit(i)
        return cls(_upload_bitmap(gpu_device, buf.data, n), gpu_device, n, NullBitBufferGpu.new(gpu_device, nulls))

    @classmethod
    def from_slice(cls, value, gpu_device: GpuDevice):
        b = BooleanBufferBuilder.from_bools(value)
        return cls(_upload_bitmap(gpu_device, b.data, b.len), gpu_device, b.len, None)

    @classmethod
    def from_bytes_slice(cls, value, gpu_device: GpuDevice):
        """NB: like the reference, `len` is the BYTE count of the slice (boolean_gpu.rs:70-81)."""
        data = np.ascontiguousarray(value, dtype=np.uint8)
        return cls(_upload_bitmap(gpu_device, data, len(data) * 8), gpu_device, len(data), None)

    def raw_bytes(self) -> np.ndarray:
        return self.gpu_device.retrive_data(self.data, (self.len + 7) // 8)

    def raw_values(self) -> np.ndarray:
        raw = self.raw_bytes()
        return np.unpackbits(raw, bitorder="little")[: self.len].astype(bool)

    def values(self) -> list:
        vals = self.raw_values()
        if self.null_buffer is None:
            return [bool(v) for v in vals]
        nulls = self.null_buffer.raw_values()
        return [bool(vals[i]) if (nulls[i // 8] >> (i % 8)) & 1 else None for i in range(self.len)]

    @classmethod
    def broadcast_op(cls, value: bool, len_: int, pipeline: ArrowComputePipeline):
        dev = pipeline.device
        out = dev.create_empty_buffer(max(bitmap_bytes(len_), 8))
        capi.call("agpu_broadcast", pipeline._handle, capi.BOOL, 1 if value else 0, C.c_void_p(out.ptr), len_)
        pipeline.keep(out)
        return cls(out, dev, len_, None)

    @classmethod
    def broadcast(cls, value: bool, len_: int, gpu_device: GpuDevice):
        p = ArrowComputePipeline(gpu_device, "broadcast")
        arr = cls.broadcast_op(value, len_, p)
        p.finish()
        return arr

    def clone_array(self):
        data = self.gpu_device.clone_buffer(self.data)
        return BooleanArrayGPU(data, self.gpu_device, self.len, NullBitBufferGpu.clone_null_bit_buffer(self.null_buffer))

    def __repr__(self):
        return f"BooleanArrayGPU(len={self.len}, nulls={'yes' if self.null_buffer else 'no'})"


def _u32_create_broadcast_buffer_op(value: int, len_: int, pipeline: ArrowComputePipeline):
    """`UInt32ArrayGPU::create_broadcast_buffer_op(value, len, pipeline) -> Buffer` [ref: crates/array/src/array/u32_gpu.rs:48-64]:
    a bare device buffer of `len` copies of `value` (the reference's helper for index / count columns), enqueued on the pipeline."""
    dev = pipeline.device
    out = dev.create_empty_buffer(max(4 * int(len_), 1))
    capi.call("agpu_broadcast", pipeline._handle, capi.U32, int(value) & 0xFFFFFFFF, C.c_void_p(out.ptr), int(len_))
    pipeline.keep(out)
    return out


def _u32_create_broadcast_buffer(value: int, len_: int, gpu_device: GpuDevice):
    """`UInt32ArrayGPU::create_broadcast_buffer(value, len, gpu_device) -> Buffer` [ref: u32_gpu.rs:36-46]: the immediate form."""
    p = ArrowComputePipeline(gpu_device, "create_broadcast_buffer")
    out = _u32_create_broadcast_buffer_op(value, len_, p)
    p.finish()
    return out


UInt32ArrayGPU.create_broadcast_buffer = staticmethod(_u32_create_broadcast_buffer)
UInt32ArrayGPU.create_broadcast_buffer_op = staticmethod(_u32_create_broadcast_buffer_op)

PRIMITIVE_TYPES = (Float32ArrayGPU, UInt32ArrayGPU, UInt16ArrayGPU, UInt8ArrayGPU, Int32ArrayGPU, Int16ArrayGPU,
                   Int8ArrayGPU, Date32ArrayGPU)
ARRAY_OF_TYPE = {t.ARROW_TYPE: t for t in PRIMITIVE_TYPES}
ARRAY_OF_TYPE[ArrowType.BooleanType] = BooleanArrayGPU


class ScalarValue:
    """`enum ScalarValue` (crates/array/src/kernels/mod.rs:7-17): ScalarValue.F32(1.0), ScalarValue.BOOL(True) …"""

    _KINDS = {"F32": Float32ArrayGPU, "U32": UInt32ArrayGPU, "U16": UInt16ArrayGPU, "U8": UInt8ArrayGPU,
              "I32": Int32ArrayGPU, "I16": Int16ArrayGPU, "I8": Int8ArrayGPU, "BOOL": BooleanArrayGPU}

    def __init__(self, kind: str, value):
        if kind not in self._KINDS:
            raise ArrowErrorGPU("ArgumentError", f"unknown scalar kind {kind}")
        self.kind = kind
        self.value = value

    def array_type(self):
        return self._KINDS[self.kind]


for _k in ScalarValue._KINDS:
    setattr(ScalarValue, _k, staticmethod(lambda v, _k=_k: ScalarValue(_k, v)))


class ScalarArray:
    """`enum ScalarArray` (crates/array/src/utils/mod.rs:2-11): host values tagged with their type — ScalarArray.F32Vec([1.0, 2.0]),
    ScalarArray.BOOLVec([True]) …; what `ArrowArrayGPU::get_raw_values` returns.  `From<Vec<T>>` is `ScalarArray.from_vec(values, dtype)`."""

    _KINDS = {"F32Vec": (np.float32, "F32"), "U32Vec": (np.uint32, "U32"), "U16Vec": (np.uint16, "U16"), "U8Vec": (np.uint8, "U8"),
              "I32Vec": (np.int32, "I32"), "I16Vec": (np.int16, "I16"), "I8Vec": (np.int8, "I8"), "BOOLVec": (np.bool_, "BOOL")}

    def __init__(self, kind: str, values):
        if kind not in self._KINDS:
            raise ArrowErrorGPU("ArgumentError", f"unknown ScalarArray variant {kind}")
        self.kind = kind
        self.values = np.asarray(values, dtype=self._KINDS[kind][0])

    @classmethod
    def from_vec(cls, values, np_dtype) -> "ScalarArray":
        for kind, (dt, _) in cls._KINDS.items():
            if np.dtype(dt) == np.dtype(np_dtype):
                return cls(kind, values)
        raise ArrowErrorGPU("ArgumentError", f"no ScalarArray variant for {np.dtype(np_dtype)}")

    @classmethod
    def of(cls, array_cls, values) -> "ScalarArray":
        """the variant an array class's raw values belong to (Date32 → I32Vec, as `Vec<i32>.into()` in the reference)"""
        if array_cls is BooleanArrayGPU:
            return cls("BOOLVec", values)
        return cls.from_vec(values, array_cls.NP_DTYPE)

    def scalar_kind(self) -> str:
        return self._KINDS[self.kind][1]

    def __len__(self):
        return len(self.values)

    def __iter__(self):
        return iter(self.values.tolist())

    def __getitem__(self, i):
        return self.values[i]

    def tolist(self):
        return self.values.tolist()

    def __eq__(self, other):  # #[derive(PartialEq)]: same variant and same values; a bare sequence compares by values
        if isinstance(other, ScalarArray):
            return self.kind == other.kind and np.array_equal(self.values, other.values, equal_nan=False)
        try:
            return bool(np.array_equal(self.values, np.asarray(other)))
        except Exception:  # noqa: BLE001
            return NotImplemented

    def __repr__(self):
        return f"ScalarArray.{self.kind}({self.values.tolist()!r})"


for _k in ScalarArray._KINDS:
    setattr(ScalarArray, _k, staticmethod(lambda v, _k=_k: ScalarArray(_k, v)))


class Operand:
    """`enum Operand { Scalar(ScalarValue), Array(ArrowArrayGPU) }` (crates/array/src/kernels/mod.rs:19-24): what a kernel combines
    an array with.  `Operand.Scalar(ScalarValue.F32(2.0))`, `Operand.Array(arr)`; `.as_array(len_hint, device)` gives the 1-element
    (scalar) or full array the `*_dyn` entry points dispatch on (a 1-element array IS the scalar form there:
    crates/arithmetic/src/arithmetic_kernels.rs:225-267)."""

    def __init__(self, kind: str, value):
        if kind == "Scalar" and not isinstance(value, ScalarValue):
            raise ArrowErrorGPU("ArgumentError", "Operand.Scalar holds a ScalarValue")
        if kind == "Array" and not isinstance(value, (PrimitiveArrayGpu, BooleanArrayGPU)):
            raise ArrowErrorGPU("ArgumentError", "Operand.Array holds an ArrowArrayGPU")
        if kind not in ("Scalar", "Array"):
            raise ArrowErrorGPU("ArgumentError", f"unknown Operand variant {kind}")
        self.kind, self.value = kind, value

    @staticmethod
    def Scalar(value: "ScalarValue") -> "Operand":  # noqa: N802 — the reference's variant names
        return Operand("Scalar", value)

    @staticmethod
    def Array(value) -> "Operand":  # noqa: N802
        return Operand("Array", value)

    def is_scalar(self) -> bool:
        return self.kind == "Scalar"

    def as_array(self, device: GpuDevice):
        """the array a `*_dyn` kernel takes for this operand: the array itself, or the scalar as a 1-element array"""
        if self.kind == "Array":
            return self.value
        cls = self.value.array_type()
        return cls.from_slice([self.value.value], device)

    def __repr__(self):
        return f"Operand.{self.kind}({self.value!r})"


def broadcast_op_dyn(value: ScalarValue, len_: int, pipeline: ArrowComputePipeline) -> ArrowArrayGPU:
    """[ref: crates/array/src/array/mod.rs:204-219]"""
    return value.array_type().broadcast_op(value.value, len_, pipeline)


def broadcast_dyn(value: ScalarValue, len_: int, device: GpuDevice) -> ArrowArrayGPU:
    """[ref: crates/array/src/array/mod.rs:189-200]"""
    return value.array_type().broadcast(value.value, len_, device)
