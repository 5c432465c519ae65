#!/usr/bin/env python3
"""Extract the reference's unit-test VECTORS (inputs + expected outputs) into tests/golden/reference_vectors.json.

Runs only in the build container, where the reference checkout is mounted at /root/reference (read-only).  It reads
the Rust test-macro invocations (`test_array_op!`, `test_scalar_op!`, `test_unary_op!`, `test_unary_op_float!`,
`test_float_array_op!`, `test_float_scalar_op!`, `test_cast_op!`, `test_bitcast_op!`, `test_sum!`, `test_take_op!`,
`test_put_op!`, `test_merge_op!`, `test_broadcast!`) and evaluates their literal argument expressions into plain
data.  No reference source text is copied: each record holds the test name, a file:line citation, the array type
names, the operation names and the numeric vectors.

Float expectations written as Rust expressions (`1.0f32.sin()`) are evaluated as float32(f(float64(x))) — the
reference only pins them to 1e-2 absolute (crates/test_macros/src/lib.rs:88-109), recorded as "tol": 0.01.

Usage: python tools/extract_golden.py [/root/reference] [tests/golden/reference_vectors.json]
"""
from __future__ import annotations

import json
import math
import os
import re
import sys

import numpy as np

MACROS = [
    "test_array_op", "test_bitcast_op", "test_broadcast", "test_cast_op", "test_float_array_op",
    "test_float_scalar_op", "test_merge_op", "test_put_op", "test_scalar_op", "test_sum", "test_take_op",
    "test_unary_op", "test_unary_op_float",
]

ARRAY_DTYPE = {
    "Float32ArrayGPU": "f32", "UInt32ArrayGPU": "u32", "UInt16ArrayGPU": "u16", "UInt8ArrayGPU": "u8",
    "Int32ArrayGPU": "i32", "Int16ArrayGPU": "i16", "Int8ArrayGPU": "i8", "Date32ArrayGPU": "date32",
    "BooleanArrayGPU": "bool",
}
NP = {"f32": np.float32, "u32": np.uint32, "u16": np.uint16, "u8": np.uint8, "i32": np.int32, "i16": np.int16,
      "i8": np.int8, "date32": np.int32}
INT_BITS = {"u32": (32, False), "u16": (16, False), "u8": (8, False), "i32": (32, True), "i16": (16, True),
            "i8": (8, True), "date32": (32, True)}


def balanced(s, i):
    depth = 0
    j = i
    while j < len(s):
        c = s[j]
        if c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
            if depth == 0:
                return j
        j += 1
    raise ValueError("unbalanced")


def split_top(s):
    out, depth, cur = [], 0, ""
    for c in s:
        if c in "([{":
            depth += 1
        if c in ")]}":
            depth -= 1
        if c == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += c
    if cur.strip():
        out.append(cur.strip())
    return out


class F32:
    """float32 value with the Rust f32 methods the tests use; math in float64, rounded once."""

    def __init__(self, v):
        self.v = np.float32(v.v if isinstance(v, F32) else v)

    def _f(self, fn):
        with np.errstate(all="ignore"):
            x = np.float64(self.v)
            return F32(np.float32(fn(x)))

    def sin(self): return self._f(np.sin)
    def cos(self): return self._f(np.cos)
    def sinh(self): return self._f(np.sinh)
    def acos(self): return self._f(np.arccos)
    def exp(self): return self._f(np.exp)
    def exp2(self): return self._f(np.exp2)
    def ln(self): return self._f(np.log)
    def log2(self): return self._f(np.log2)
    def sqrt(self): return self._f(np.sqrt)
    def cbrt(self): return self._f(np.cbrt)
    def abs(self): return F32(abs(self.v))
    def powf(self, y):
        y = y.v if isinstance(y, F32) else y
        with np.errstate(all="ignore"):
            return F32(np.float32(np.power(np.float64(self.v), np.float64(y))))
    def __neg__(self): return F32(-self.v)
    def __float__(self): return float(self.v)


def wrap_int(v, dtype):
    bits, signed = INT_BITS[dtype]
    v = int(v) & ((1 << bits) - 1)
    if signed and v >= 1 << (bits - 1):
        v -= 1 << bits
    return v


CONSTS = {
    "i32::MAX": 2**31 - 1, "i32::MIN": -(2**31), "u32::MAX": 2**32 - 1, "u32::MIN": 0,
    "i16::MAX": 2**15 - 1, "i16::MIN": -(2**15), "u16::MAX": 2**16 - 1,
    "i8::MAX": 127, "i8::MIN": -128, "u8::MAX": 255,
    "f32::NAN": "F32(float('nan'))", "f32::INFINITY": "F32(float('inf'))", "f32::NEG_INFINITY": "F32(float('-inf'))",
}



def replace_as(e: str) -> str:
    """`<operand> as <type>` → CAST('<type>', <operand>) / F32(<operand>); operand = balanced (...) group or a token."""
    while True:
        m = re.search(r"\s+as\s+(f32|u8|u16|u32|i8|i16|i32)\b", e)
        if not m:
            return e
        end = m.start()
        if e[end - 1] == ")":
            depth, k = 0, end - 1
            while k >= 0:
                if e[k] == ")":
                    depth += 1
                elif e[k] == "(":
                    depth -= 1
                    if depth == 0:
                        break
                k -= 1
            # include a call prefix such as FROM_BITS( or F32(
            while k > 0 and (e[k - 1].isalnum() or e[k - 1] == "_"):
                k -= 1
            start = k
        else:
            k = end
            while k > 0 and (e[k - 1].isalnum() or e[k - 1] in "_.:"):
                k -= 1
            start = k
        operand = e[start:end]
        t = m.group(1)
        rep = f"F32({operand})" if t == "f32" else f"CAST('{t}', {operand})"
        e = e[:start] + rep + e[m.end():]


def rust_to_py(e: str) -> str:
    e = re.sub(r"//[^\n]*", "", e)          # comments
    e = e.replace("vec!", "")
    for k, v in CONSTS.items():
        e = e.replace(k, f"({v})")
    e = re.sub(r"f32::from_bits\(([^()]*(?:\([^()]*\))?[^()]*)\)", r"FROM_BITS(\1)", e)
    e = replace_as(e)
    # typed literals
    e = re.sub(r"(?<![\w.])(\d+\.\d+|\d+)_?f32", r"F32(\1)", e)
    e = re.sub(r"(?<![\w.])(-?\d+)_?(u8|u16|u32|i8|i16|i32)\b", r"\1", e)
    e = re.sub(r"Some\(", "(", e)
    e = e.replace("None", "None").replace("true", "True").replace("false", "False")
    e = re.sub(r"!(?=[\w(-])", "~", e)       # bitwise not
    e = re.sub(r"\[\s*([^\[\];]+);\s*(\d+)\s*\]", r"([\1] * \2)", e)  # [x; n]
    return e


def FROM_BITS(u):
    return F32(np.array([int(u) & 0xFFFFFFFF], dtype=np.uint32).view(np.float32)[0])


def CAST(t, v):
    if isinstance(v, F32):
        v = int(v.v)
    return wrap_int(v, t)


def ev(e: str):
    return eval(rust_to_py(e), {"F32": F32, "FROM_BITS": FROM_BITS, "CAST": CAST, "__builtins__": {"float": float}})


def enc_scalar(v, dtype):
    if v is None:
        return None
    if dtype == "bool":
        return bool(v)
    if dtype == "f32":
        f = np.float32(v.v if isinstance(v, F32) else v)
        if np.isnan(f):
            return "nan"
        if np.isinf(f):
            return "inf" if f > 0 else "-inf"
        return float(f)
    if isinstance(v, F32):
        v = int(v.v)
    return wrap_int(v, dtype)


def enc(v, dtype):
    if isinstance(v, (list, tuple)):
        return [enc_scalar(x, dtype) for x in v]
    return enc_scalar(v, dtype)


def bits_of(v):
    f = np.float32(v.v if isinstance(v, F32) else v)
    return int(np.array([f], dtype=np.float32).view(np.uint32)[0])


def strip_comments(s: str) -> str:
    """Blank out /* */ and // comments, keeping newlines so line numbers stay right."""
    def blank(m):
        return re.sub(r"[^\n]", " ", m.group(0))
    s = re.sub(r"/\*.*?\*/", blank, s, flags=re.S)
    return re.sub(r"//[^\n]*", blank, s)


def scan(root):
    recs = []
    for dp, _, fns in sorted(os.walk(os.path.join(root, "crates"))):
        for fn in sorted(fns):
            if not fn.endswith(".rs"):
                continue
            path = os.path.join(dp, fn)
            s = strip_comments(open(path).read())
            for m in re.finditer(r"\b(" + "|".join(MACROS) + r")!\(", s):
                if "macro_rules!" in s[max(0, m.start() - 20): m.start()]:
                    continue
                i = m.end() - 1
                j = balanced(s, i)
                body = s[i + 1: j]
                ignored = "ignore" in body.split("test_")[0] if body.lstrip().startswith("#[") else False
                b = body
                while b.lstrip().startswith("#["):
                    b = b.lstrip()
                    k = balanced(b, 1)
                    b = b[k + 1:]
                args = split_top(b)
                line = s.count("\n", 0, m.start()) + 1
                recs.append((os.path.relpath(path, root), line, m.group(1), args, ignored))
    return recs


def is_ident(x):
    return re.fullmatch(r"[A-Za-z_][A-Za-z_0-9]*", x) is not None


def convert(rel, line, macro, args, ignored):
    r = {"name": args[0], "ref": f"{rel}:{line}", "macro": macro}
    if ignored:
        r["ignored_in_reference"] = True
    a = args[1:]
    if macro in ("test_array_op", "test_float_array_op"):
        t1, t2, to, op = a[0], a[1], a[2], a[3]
        rest = a[4:]
        op_dyn = None
        if is_ident(rest[0]):
            op_dyn, rest = rest[0], rest[1:]
        d1, d2, do = ARRAY_DTYPE[t1], ARRAY_DTYPE[t2], ARRAY_DTYPE[to]
        r.update(kind="array_op", op=op, op_dyn=op_dyn, types=[t1, t2, to],
                 a=enc(ev(rest[0]), d1), b=enc(ev(rest[1]), d2), expected=enc(ev(rest[2]), do))
        if macro == "test_float_array_op":
            r["tol"] = 0.01
    elif macro in ("test_scalar_op", "test_float_scalar_op"):
        ti, ts, to, inp, op, op_dyn, sc, out = a
        r.update(kind="scalar_op", op=op, op_dyn=op_dyn, types=[ti, ts, to],
                 a=enc(ev(inp), ARRAY_DTYPE[ti]), scalar=enc(ev(sc), ARRAY_DTYPE[ts]),
                 expected=enc(ev(out), ARRAY_DTYPE[to]))
        if macro == "test_float_scalar_op":
            r["tol"] = 0.01
    elif macro in ("test_unary_op", "test_unary_op_float"):
        ti, to, inp, op = a[0], a[1], a[2], a[3]
        rest = a[4:]
        op_dyn = None
        if len(rest) == 2:
            op_dyn, rest = rest[0], rest[1:]
        r.update(kind="unary_op", op=op, op_dyn=op_dyn, types=[ti, to],
                 a=enc(ev(inp), ARRAY_DTYPE[ti]), expected=enc(ev(rest[0]), ARRAY_DTYPE[to]))
        if macro == "test_unary_op_float":
            r["tol"] = 0.01
    elif macro in ("test_cast_op", "test_bitcast_op"):
        ti, to, inp, cast_type, out = a
        r.update(kind="cast" if macro == "test_cast_op" else "bitcast", types=[ti, to], cast_type=cast_type,
                 a=enc(ev(inp), ARRAY_DTYPE[ti]))
        if macro == "test_bitcast_op":
            r["expected_bits"] = [bits_of(x) for x in ev(out)]
        else:
            r["expected"] = enc(ev(out), ARRAY_DTYPE[to])
    elif macro == "test_sum":
        ty, base, size, out = a
        d = ARRAY_DTYPE[ty]
        r.update(kind="sum", types=[ty], base=enc(ev(base), d), size=int(ev(size)), expected=enc(ev(out), d))
    elif macro == "test_broadcast":
        ty, inp = a
        r.update(kind="broadcast", types=[ty], value=enc(ev(inp), ARRAY_DTYPE[ty]), size=100)
    elif macro == "test_take_op":
        t1, t2, to, op = a[0], a[1], a[2], a[3]
        rest = a[4:]
        op_dyn = None
        if is_ident(rest[0]):
            op_dyn, rest = rest[0], rest[1:]
        r.update(kind="take", op=op, op_dyn=op_dyn, types=[t1, t2, to],
                 values=enc(ev(rest[0]), ARRAY_DTYPE[t1]), indexes=enc(ev(rest[1]), "u32"),
                 expected=enc(ev(rest[2]), ARRAY_DTYPE[to]))
    elif macro == "test_put_op":
        ty, op = a[0], a[1]
        rest = a[2:]
        op_dyn = None
        if is_ident(rest[0]):
            op_dyn, rest = rest[0], rest[1:]
        d = ARRAY_DTYPE[ty]
        r.update(kind="put", op=op, op_dyn=op_dyn, types=[ty], src=enc(ev(rest[0]), d), dst=enc(ev(rest[1]), d),
                 src_indexes=enc(ev(rest[2]), "u32"), dst_indexes=enc(ev(rest[3]), "u32"), expected=enc(ev(rest[4]), d))
    elif macro == "test_merge_op":
        t1, t2, to, op = a[0], a[1], a[2], a[3]
        rest = a[4:]
        op_dyn = None
        if is_ident(rest[0]):
            op_dyn, rest = rest[0], rest[1:]
        r.update(kind="merge", op=op, op_dyn=op_dyn, types=[t1, t2, to], a=enc(ev(rest[0]), ARRAY_DTYPE[t1]),
                 b=enc(ev(rest[1]), ARRAY_DTYPE[t2]), mask=enc(ev(rest[2]), "bool"),
                 expected=enc(ev(rest[3]), ARRAY_DTYPE[to]))
    else:
        raise ValueError(macro)
    return r


def main():
    root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(__file__), "..", "tests", "golden",
                                                             "reference_vectors.json")
    recs = [convert(*x) for x in scan(root)]
    recs.sort(key=lambda r: r["ref"])
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        f.write("[\n" + ",\n".join(json.dumps(r) for r in recs) + "\n]\n")
    kinds = {}
    for r in recs:
        kinds[r["kind"]] = kinds.get(r["kind"], 0) + 1
    print(f"wrote {len(recs)} vectors to {out}: {kinds}")


if __name__ == "__main__":
    main()
