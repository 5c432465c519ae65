"""Oracle array model — TEST INFRASTRUCTURE ONLY.

The same public surface as `arrow_gpu_amd` (array classes, trait methods, `*_dyn` functions) but computed on the
host with numpy + oracle/agpu_oracle.c.  It restates the reference's HOST-SIDE rules independently of the product
package — validity merge (crates/array/src/array/null_bit_buffer.rs:168-243), array-vs-scalar dispatch by len
(crates/arithmetic/src/arithmetic_kernels.rs:101-119), dyn type tables, output types — so that tests/ can run one
fixture runner against both namespaces: this one on CPU (pins the oracle on the reference's golden vectors) and the
HIP one on the GPU.  Nothing here is imported by the product.
"""
from __future__ import annotations

import numpy as np

import oracle as O


class OracleUnsupported(RuntimeError):
    """Stands for the reference's `panic!("Operation … not supported …")`."""


OperationNotSupported = OracleUnsupported
CastingNotSupported = OracleUnsupported
ArrowErrorGPU = RuntimeError


class GpuDevice:  # placeholder so the fixture runner can pass a "device" around
    def __init__(self, ordinal: int = 0):
        self.ordinal = ordinal


class ArrowComputePipeline:
    def __init__(self, device, label=None):
        self.device = device

    def finish(self):
        pass


_DEV = GpuDevice()


def GPU_DEVICE():
    return _DEV


class ArrowArrayGPU:
    DTYPE = None

    def __init__(self, data, len_, validity):
        self.data = data          # np array of values, or uint8 bitmap for Boolean
        self.len = len_
        self.validity = validity  # uint8 bitmap (whole 64-bit words) or None
        self.gpu_device = _DEV

    @property
    def null_buffer(self):
        return self.validity

    def get_gpu_device(self):
        return _DEV

    def into(self):
        return self

    @classmethod
    def try_from(cls, v):
        if type(v) is cls:
            return v
        raise OracleUnsupported(f"could not cast {type(v).__name__} into {cls.__name__}")

    def _validity_list(self):
        if self.validity is None:
            return [True] * self.len
        return list(O.unpack_bits(self.validity, self.len))


class PrimitiveArrayGpu(ArrowArrayGPU):
    NP = None

    @classmethod
    def from_optional_slice(cls, values, device=None):
        data, validity = O.from_optional(values, cls.DTYPE)
        return cls(data, len(values), validity)

    @classmethod
    def from_slice(cls, values, device=None):
        if isinstance(values, np.ndarray):
            return cls(np.ascontiguousarray(values, dtype=cls.NP), len(values), None)
        arr = np.array([_wrap(v, cls.NP) for v in values], dtype=cls.NP)
        return cls(arr, len(values), None)

    @classmethod
    def broadcast(cls, value, n, device=None):
        return cls(O.broadcast(cls.DTYPE, _wrap(value, cls.NP), n), n, None)

    def raw_values(self):
        return self.data[: self.len].copy()

    def values(self):
        v = self._validity_list()
        return [self.data[i].item() if v[i] else None for i in range(self.len)]


def _wrap(v, np_dtype):
    if np.issubdtype(np_dtype, np.integer):
        info = np.iinfo(np_dtype)
        v = int(v)
        if not (info.min <= v <= info.max):
            v = (v - info.min) % (1 << info.bits) + info.min
    return np_dtype(v)


def _prim(name, code):
    return type(name, (PrimitiveArrayGpu,), {"DTYPE": code, "NP": O.NP_DTYPE[code], "ITEM_SIZE": np.dtype(O.NP_DTYPE[code]).itemsize})


Float32ArrayGPU = _prim("Float32ArrayGPU", O.F32)
UInt32ArrayGPU = _prim("UInt32ArrayGPU", O.U32)
UInt16ArrayGPU = _prim("UInt16ArrayGPU", O.U16)
UInt8ArrayGPU = _prim("UInt8ArrayGPU", O.U8)
Int32ArrayGPU = _prim("Int32ArrayGPU", O.I32)
Int16ArrayGPU = _prim("Int16ArrayGPU", O.I16)
Int8ArrayGPU = _prim("Int8ArrayGPU", O.I8)
Date32ArrayGPU = _prim("Date32ArrayGPU", O.DATE32)


class BooleanArrayGPU(ArrowArrayGPU):
    DTYPE = O.BOOL

    @classmethod
    def from_optional_slice(cls, values, device=None):
        data, validity = O.from_optional(values, O.BOOL)
        return cls(data, len(values), validity)

    @classmethod
    def from_slice(cls, values, device=None):
        return cls(O.pack_bits(values), len(values), None)

    @classmethod
    def broadcast(cls, value, n, device=None):
        return cls(O.broadcast(O.BOOL, value, n), n, None)

    def raw_values(self):
        return O.unpack_bits(self.data, self.len)

    def values(self):
        v = self._validity_list()
        r = self.raw_values()
        return [bool(r[i]) if v[i] else None for i in range(self.len)]

    def any(self):
        return O.bitmap_any(self.data, self.len)

    def all(self):
        return O.bitmap_popcount(self.data, self.len) == self.len


class ArrowType:
    pass


_TYPES = {"BooleanType": BooleanArrayGPU, "Float32Type": Float32ArrayGPU, "UInt32Type": UInt32ArrayGPU,
          "UInt16Type": UInt16ArrayGPU, "UInt8Type": UInt8ArrayGPU, "Int32Type": Int32ArrayGPU,
          "Int16Type": Int16ArrayGPU, "Int8Type": Int8ArrayGPU, "Date32Type": Date32ArrayGPU}
for _n, _c in _TYPES.items():
    setattr(ArrowType, _n, _c)

# ------------------------------------------------------------------ helpers


def _merge_validity(a, b):
    return O.validity_and(a.validity, b.validity, a.len)


def _clone(v):
    return None if v is None else v.copy()


def _attach(classes, name, fn):
    for c in classes:
        setattr(c, name + "_op", lambda self, *args, _fn=fn: _fn(self, *args[:-1]))
        setattr(c, name, fn)


def _array_op(op, out_cls=None):
    def fn(self, other):
        assert self.len == other.len
        cls = out_cls or type(self)
        if isinstance(self, BooleanArrayGPU):
            data = O.bitmap_binary(op, self.data, other.data, self.len)
        else:
            data = O.binary(op, self.DTYPE, self.data, other.data)
        return cls(data, self.len, _merge_validity(self, other))
    return fn


def _scalar_op(op):
    def fn(self, scalar):
        return type(self)(O.scalar(op, self.DTYPE, self.data, scalar.data[:1]), self.len, _clone(self.validity))
    return fn


def _unary_op(op, out_cls=None):
    def fn(self):
        cls = out_cls or type(self)
        if isinstance(self, BooleanArrayGPU):
            return cls(O.bitmap_not(self.data, self.len), self.len, _clone(self.validity))
        return cls(O.unary(op, self.DTYPE, self.data), self.len, _clone(self.validity))
    return fn


def _cmp_op(op):
    def fn(self, other):
        assert self.len == other.len
        return BooleanArrayGPU(O.compare(op, self.DTYPE, self.data, other.data), self.len, _merge_validity(self, other))
    return fn


_F32 = (Float32ArrayGPU,)
_I32ish = (Int32ArrayGPU, Date32ArrayGPU, UInt32ArrayGPU)
_ALL = (Float32ArrayGPU, UInt32ArrayGPU, UInt16ArrayGPU, UInt8ArrayGPU, Int32ArrayGPU, Int16ArrayGPU, Int8ArrayGPU, Date32ArrayGPU)
_INTS = (Int32ArrayGPU, UInt32ArrayGPU, UInt16ArrayGPU, Int16ArrayGPU, UInt8ArrayGPU, Int8ArrayGPU)
_TRIG = (Float32ArrayGPU, UInt16ArrayGPU, UInt8ArrayGPU, Int16ArrayGPU, Int8ArrayGPU)

# arithmetic
_attach(_F32 + _I32ish + (UInt16ArrayGPU,), "add_scalar", _scalar_op(O.OP_ADD))
for _n, _o in (("sub_scalar", O.OP_SUB), ("mul_scalar", O.OP_MUL), ("div_scalar", O.OP_DIV), ("rem_scalar", O.OP_REM)):
    _attach(_F32 + _I32ish, _n, _scalar_op(_o))
for _n, _o in (("add", O.OP_ADD), ("sub", O.OP_SUB), ("mul", O.OP_MUL)):
    _attach(_F32 + _I32ish, _n, _array_op(_o))
_attach(_F32, "div", _array_op(O.OP_DIV))
_attach(_F32, "neg", _unary_op(O.UN_NEG))


def _sum(self):
    return type(self)(np.array([O.reduce(O.RED_SUM, self.DTYPE, self.data)], dtype=self.NP), 1, None)


_attach((Float32ArrayGPU, Int32ArrayGPU, UInt32ArrayGPU), "sum", _sum)

# compare
for _n, _o in (("gt", O.CMP_GT), ("gteq", O.CMP_GTEQ), ("lt", O.CMP_LT), ("lteq", O.CMP_LTEQ), ("eq", O.CMP_EQ)):
    _attach(_ALL, _n, _cmp_op(_o))
_attach(_ALL, "max", _array_op(O.OP_MAX))
_attach(_ALL, "min", _array_op(O.OP_MIN))

# logical
for _n, _o in (("bitwise_and", O.OP_AND), ("bitwise_or", O.OP_OR), ("bitwise_xor", O.OP_XOR)):
    _attach(_INTS + (BooleanArrayGPU,), _n, _array_op(_o))
_attach(_INTS + (BooleanArrayGPU,), "bitwise_not", _unary_op(O.UN_NOT))
_attach(_INTS, "bitwise_shl", _array_op(O.OP_SHL))
_attach(_INTS, "bitwise_shr", _array_op(O.OP_SHR))

# math / trig
_attach((Float32ArrayGPU, Int32ArrayGPU), "abs", _unary_op(O.UN_ABS))
for _n, _o in (("sqrt", O.UN_SQRT), ("cbrt", O.UN_CBRT), ("exp", O.UN_EXP), ("exp2", O.UN_EXP2), ("log", O.UN_LOG), ("log2", O.UN_LOG2)):
    _attach(_F32, _n, _unary_op(_o))
_attach((Float32ArrayGPU, Int32ArrayGPU), "power", _array_op(O.OP_POW))
for _n, _o in (("sin", O.UN_SIN), ("cos", O.UN_COS), ("sinh", O.UN_SINH)):
    _attach(_TRIG, _n, _unary_op(_o, Float32ArrayGPU))
_attach(_F32, "acos", _unary_op(O.UN_ACOS, Float32ArrayGPU))

# cast
CAST_TABLE = {
    Int8ArrayGPU: (UInt8ArrayGPU, UInt16ArrayGPU, UInt32ArrayGPU, Int16ArrayGPU, Int32ArrayGPU, Float32ArrayGPU),
    Int16ArrayGPU: (Int32ArrayGPU, UInt16ArrayGPU, UInt32ArrayGPU, Float32ArrayGPU),
    UInt8ArrayGPU: (UInt16ArrayGPU, UInt32ArrayGPU, Int8ArrayGPU, Int16ArrayGPU, Int32ArrayGPU, Float32ArrayGPU),
    UInt16ArrayGPU: (UInt32ArrayGPU, Int16ArrayGPU, Int32ArrayGPU, Float32ArrayGPU),
    # f32 → u8 is the reference's only narrowing cast [cast/src/f32_cast.rs:8-31]; the other five are REFERENCE-ABSENT
    # (north_star "i8/i16/u8/u16 <-> f32"), defined by analogy with cast_u8.wgsl — see include/arrow_gpu.h agpu_cast
    Float32ArrayGPU: (UInt8ArrayGPU, Int8ArrayGPU, Int16ArrayGPU, UInt16ArrayGPU, Int32ArrayGPU, UInt32ArrayGPU),
    BooleanArrayGPU: (Float32ArrayGPU,),
}


def _cast(self, into):
    if into not in CAST_TABLE.get(type(self), ()):
        raise OracleUnsupported("Casting not supported")
    if isinstance(self, BooleanArrayGPU):
        data = O.cast(O.BOOL, into.DTYPE, self.data, self.len)
    elif np.dtype(self.NP).itemsize == np.dtype(into.NP).itemsize and into is not Float32ArrayGPU and type(self) is not Float32ArrayGPU:
        data = self.data.view(into.NP).copy()  # sign reinterpret = buffer copy (cast/src/lib.rs:69-86)
    else:
        data = O.cast(self.DTYPE, into.DTYPE, self.data)
    return into(data, self.len, _clone(self.validity))


def _bitcast(self, into):
    if not (type(self) is UInt32ArrayGPU and into is Float32ArrayGPU):
        raise OracleUnsupported("Casting not supported")
    return into(O.bitcast(self.DTYPE, into.DTYPE, self.data), self.len, _clone(self.validity))


for _c in CAST_TABLE:
    _c.cast = _cast
    _c.cast_op = lambda self, into, pipeline: _cast(self, into)
UInt32ArrayGPU.bitcast = _bitcast
UInt32ArrayGPU.bitcast_op = lambda self, into, pipeline: _bitcast(self, into)


def cast_dyn(a, into):
    if type(a) not in CAST_TABLE:  # the reference's dyn match falls through to its panic arm
        raise OracleUnsupported("Casting not supported")
    return a.cast(into)


def bitcast_dyn(a, into):
    if type(a) is not UInt32ArrayGPU:
        raise OracleUnsupported("Casting not supported")
    return a.bitcast(into)


# routines
def _merge(self, other, mask):
    assert self.len == other.len == mask.len
    if isinstance(self, BooleanArrayGPU):
        data = O.merge_bits(self.data, other.data, mask.data, self.len)
    else:
        data = O.merge(self.ITEM_SIZE, self.data, other.data, mask.data)
    if self.validity is None and other.validity is None and mask.validity is None:
        v = None
    else:
        v = O.merge_validity(self.validity, other.validity, mask.data, mask.validity, self.len)
    return type(self)(data, self.len, v)


def _take(self, indexes):
    idx = indexes.data
    if isinstance(self, BooleanArrayGPU):
        data = O.take_bits(self.data, self.len, idx)
    else:
        data = O.take(self.ITEM_SIZE, self.data, idx)
    v = None if self.validity is None else O.take_bits(self.validity, self.len, idx)
    return type(self)(data, len(idx), v)


def _put(self, src_indexes, dst, dst_indexes):
    # null buffers are `todo!()` in the reference (routines/src/lib.rs:164-169); the build's extension:
    # the validity bit travels with the value, an absent bitmap counts as all-valid
    if self.validity is not None or dst.validity is not None:
        ones = lambda n: np.full((n + 7) // 8, 0xFF, np.uint8)  # noqa: E731
        sv = self.validity if self.validity is not None else ones(self.len)
        dv = dst.validity if dst.validity is not None else ones(dst.len)
        dst.validity = O.put_bits(sv, src_indexes.data, dv, dst_indexes.data)
    if isinstance(self, BooleanArrayGPU):
        dst.data = O.put_bits(self.data, src_indexes.data, dst.data, dst_indexes.data)
    else:
        dst.data = O.put(self.ITEM_SIZE, self.data, src_indexes.data, dst.data, dst_indexes.data)


for _c in _ALL + (BooleanArrayGPU,):
    _c.merge = _merge
    _c.merge_op = lambda self, other, mask, pipeline: _merge(self, other, mask)
    _c.take = _take
    _c.take_op = lambda self, idx, pipeline: _take(self, idx)
    _c.put = _put
    _c.put_op = lambda self, si, dst, di, pipeline: _put(self, si, dst, di)


# ------------------------------------------------------------------ dyn tables (same lists as the reference)
def _dyn2(name, method, same, mixed=()):
    def fn(a, b):
        if (type(a) is type(b) and type(a) in same) or (type(a), type(b)) in mixed:
            return getattr(a, method)(b)
        raise OracleUnsupported(f"Operation {name} not supported for type {type(a).__name__} {type(b).__name__}")
    return fn


def _dyn1(name, method, types):
    def fn(a):
        if type(a) in types:
            return getattr(a, method)()
        raise OracleUnsupported(f"Operation {name} not supported for type {type(a).__name__}")
    return fn


_mix = ((Int32ArrayGPU, Date32ArrayGPU), (Date32ArrayGPU, Int32ArrayGPU))
add_scalar_dyn = _dyn2("add_scalar_dyn", "add_scalar", (Float32ArrayGPU, Int32ArrayGPU, Date32ArrayGPU, UInt32ArrayGPU, UInt16ArrayGPU))
sub_scalar_dyn = _dyn2("sub_scalar_dyn", "sub_scalar", (Float32ArrayGPU, Int32ArrayGPU, UInt32ArrayGPU))
mul_scalar_dyn = _dyn2("mul_scalar_dyn", "mul_scalar", (Float32ArrayGPU, Int32ArrayGPU, UInt32ArrayGPU))
div_scalar_dyn = _dyn2("div_scalar_dyn", "div_scalar", (Float32ArrayGPU, Int32ArrayGPU, UInt32ArrayGPU))
rem_scalar_dyn = _dyn2("rem_scalar_dyn", "rem_scalar", (Float32ArrayGPU, Int32ArrayGPU, UInt32ArrayGPU, Date32ArrayGPU), _mix)
add_array_dyn = _dyn2("add_array_dyn", "add", (Float32ArrayGPU, UInt32ArrayGPU, Int32ArrayGPU, Date32ArrayGPU), _mix)
sub_array_dyn = _dyn2("sub_array_dyn", "sub", _F32)
mul_array_dyn = _dyn2("mul_array_dyn", "mul", _F32)
div_array_dyn = _dyn2("div_array_dyn", "div", _F32)


def _len_dispatch(array_fn, scalar_fn):
    def fn(a, b):
        x, y = a.len, b.len
        if (x == 1 and y == 1) or (x != 1 and y != 1):
            return array_fn(a, b)
        if y == 1:
            return scalar_fn(a, b)
        return scalar_fn(b, a)
    return fn


add_dyn = _len_dispatch(add_array_dyn, add_scalar_dyn)
sub_dyn = _len_dispatch(sub_array_dyn, sub_scalar_dyn)
mul_dyn = _len_dispatch(mul_array_dyn, mul_scalar_dyn)
div_dyn = _len_dispatch(div_array_dyn, div_scalar_dyn)
neg_dyn = _dyn1("neg_dyn", "neg", _F32)

gt_dyn = _dyn2("gt_dyn", "gt", _ALL)
gteq_dyn = _dyn2("gteq_dyn", "gteq", _ALL)
lt_dyn = _dyn2("lt_dyn", "lt", _ALL)
lteq_dyn = _dyn2("lteq_dyn", "lteq", _ALL)
eq_dyn = _dyn2("eq_dyn", "eq", _ALL)
max_dyn = _dyn2("max_dyn", "max", _ALL)
min_dyn = _dyn2("min_dyn", "min", _ALL)

_LOG = _INTS + (BooleanArrayGPU,)
bitwise_and_dyn = _dyn2("bitwise_and_dyn", "bitwise_and", _LOG)
bitwise_or_dyn = _dyn2("bitwise_or_dyn", "bitwise_or", _LOG)
bitwise_xor_dyn = _dyn2("bitwise_xor_dyn", "bitwise_xor", _LOG)
bitwise_not_dyn = _dyn1("bitwise_not_dyn", "bitwise_not", _LOG)
_SH = tuple((t, UInt32ArrayGPU) for t in _INTS if t is not UInt32ArrayGPU)
bitwise_shl_dyn = _dyn2("bitwise_shl_dyn", "bitwise_shl", (UInt32ArrayGPU,), _SH)
bitwise_shr_dyn = _dyn2("bitwise_shr_dyn", "bitwise_shr", (UInt32ArrayGPU,), _SH)

abs_dyn = _dyn1("abs_dyn", "abs", (Float32ArrayGPU, Int32ArrayGPU))
sqrt_dyn = _dyn1("sqrt_dyn", "sqrt", _F32)
cbrt_dyn = _dyn1("cbrt_dyn", "cbrt", _F32)
exp_dyn = _dyn1("exp_dyn", "exp", _F32)
exp2_dyn = _dyn1("exp2_dyn", "exp2", _F32)
log_dyn = _dyn1("log_dyn", "log", _F32)
log2_dyn = _dyn1("log2_dyn", "log2", _F32)
power_dyn = _dyn2("power_dyn", "power", (Int32ArrayGPU, Float32ArrayGPU))
sin_dyn = _dyn1("sin_dyn", "sin", _TRIG)
cos_dyn = _dyn1("cos_dyn", "cos", _TRIG)
sinh_dyn = _dyn1("sinh_dyn", "sinh", _TRIG)
acos_dyn = _dyn1("acos_dyn", "acos", _F32)


def merge_dyn(a, b, mask):
    if type(a) is type(b):
        return a.merge(b, mask)
    raise OracleUnsupported("merge_dyn")


def take_dyn(a, idx):
    if type(a) in (Date32ArrayGPU, UInt32ArrayGPU, Int32ArrayGPU, Float32ArrayGPU, BooleanArrayGPU):
        return a.take(idx)
    raise OracleUnsupported("take_dyn")


def put_dyn(src, si, dst, di):
    if type(src) is type(dst) and type(src) in (Float32ArrayGPU, Int32ArrayGPU, UInt32ArrayGPU, Date32ArrayGPU, BooleanArrayGPU):
        return src.put(si, dst, di)
    raise OracleUnsupported("put_dyn")


def _op_twin(fn):
    def fn_op(*args):
        return fn(*args[:-1])
    return fn_op


for _name in [n for n in list(globals()) if n.endswith("_dyn")]:
    globals()[_name.replace("_dyn", "_op_dyn")] = _op_twin(globals()[_name])
