#!/usr/bin/env python3
"""Every call site of the hand-written `global_store_dwordx4 … sc1 nt` store (csrc/common.hpp st_vec_sc1) run once:
  * ew_kernel<T, Op, MODE_UNARY | MODE_SCALAR> for 4-byte T — every unary and scalar op of f32 / i32 / u32 / Date32
  * cvt_wide_kernel ×2 widenings — i16 / u16 → f32 / i32 / u32
at sizes whose full tiles put DISTINCT data into every lane and every component of the 16-byte vector (the store-data
hazard the inline asm pads for clobbered lanes 12–15 of every 16, components 2–3), plus ragged tails.  Prints one JSON
line {site: sha256 of the output bytes}.  tests/test_gpu_sc1.py compares the default build with the AGPU_USE_SC1=0 build
(AGPU_LIB=…/libarrow_gpu_hip_nosc1.so: plain __builtin_nontemporal_store) and, in-process, with the oracle."""
from __future__ import annotations

import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice  # noqa: E402

SIZES = (64 * 4 * 7, 1_048_576 + 64 * 4 * 3 + 5, 3_000_001)

UNARY = {capi.F32: ("NEG ABS SQRT CBRT EXP EXP2 LOG LOG2 SIN COS ACOS SINH", np.float32),
         capi.I32: ("NEG ABS NOT POPCOUNT", np.int32), capi.U32: ("NOT POPCOUNT", np.uint32)}
SCALAR = {capi.F32: ("ADD SUB MUL DIV REM MIN MAX POW", np.float32),
          capi.I32: ("ADD SUB MUL DIV REM MIN MAX AND OR XOR SHL SHR POW", np.int32),
          capi.U32: ("ADD SUB MUL DIV REM MIN MAX AND OR XOR SHL SHR", np.uint32),
          capi.DATE32: ("ADD SUB MUL DIV REM", np.int32)}
WIDEN = [(capi.I16, capi.F32), (capi.U16, capi.F32), (capi.I16, capi.I32), (capi.U16, capi.U32), (capi.I16, capi.U32), (capi.U16, capi.I32)]
NP = {capi.F32: np.float32, capi.I32: np.int32, capi.U32: np.uint32, capi.I16: np.int16, capi.U16: np.uint16, capi.DATE32: np.int32}
DT_NAME = {capi.F32: "f32", capi.I32: "i32", capi.U32: "u32", capi.I16: "i16", capi.U16: "u16", capi.DATE32: "date32"}


def column(npd, n, seed):
    """distinct in every lane and component: a counter mixed with a hash, no two neighbours alike"""
    i = np.arange(n, dtype=np.uint64)
    h = (i * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)) ^ (i >> np.uint64(7))
    if npd == np.float32:
        return ((h >> np.uint64(40)).astype(np.float64) / (1 << 24) * 40.0 - 20.0).astype(np.float32)
    if np.dtype(npd).itemsize == 4:
        return (h >> np.uint64(20)).astype(np.uint32).view(npd)
    return (h >> np.uint64(30)).astype(np.uint16).view(npd)


def sites(run):
    """run(name, kind, args…) for every call site × size; yields (site, inputs, launch-spec)"""
    for n in SIZES:
        for dt, (ops, npd) in UNARY.items():
            for op in ops.split():
                yield f"unary/{DT_NAME[dt]}/{op}/{n}", ("unary", getattr(capi, "UN_" + op), dt, npd, n)
        for dt, (ops, npd) in SCALAR.items():
            for op in ops.split():
                yield f"scalar/{DT_NAME[dt]}/{op}/{n}", ("scalar", getattr(capi, "OP_" + op), dt, npd, n)
        for frm, to in WIDEN:
            yield f"cast/{DT_NAME[frm]}->{DT_NAME[to]}/{n}", ("cast", frm, to, None, n)


def main(check_oracle=False):
    dev = GpuDevice(0)
    p = ArrowComputePipeline(dev, "sc1")
    h = p._handle
    vp = lambda b: C.c_void_p(b.ptr)  # noqa: E731
    out = {}
    bad = []
    O = None
    if check_oracle:
        import oracle as O  # noqa: N812
    cache = {}
    for name, spec in sites(None):
        kind, n = spec[0], spec[4]
        if kind == "cast":
            frm, to = spec[1], spec[2]
            src = column(NP[frm], n, 11)
            key = ("col", frm, n)
            if key not in cache:
                cache[key] = dev.create_gpu_buffer_with_data(src)
            dout = dev.create_empty_buffer(4 * n + 64)
            capi.call("agpu_memset", h, vp(dout), 0xA5, 4 * n + 64)
            capi.call("agpu_cast", h, frm, to, vp(cache[key]), vp(dout), n)
            got = dev.retrive_data(dout, 4 * n + 64, pipeline=p)
            if O is not None and not np.array_equal(got[: 4 * n], np.ascontiguousarray(O.cast(frm, to, src)).view(np.uint8)):
                bad.append(name)
        else:
            op, dt, npd = spec[1], spec[2], spec[3]
            src = column(npd, n, 5 + dt)
            key = ("col", dt, n)
            if key not in cache:
                cache[key] = dev.create_gpu_buffer_with_data(src)
            dout = dev.create_empty_buffer(4 * n + 64)
            capi.call("agpu_memset", h, vp(dout), 0xA5, 4 * n + 64)
            if kind == "unary":
                capi.call("agpu_unary", h, op, dt, vp(cache[key]), vp(dout), n)
            else:
                sval = np.array([3], npd) if npd != np.float32 else np.array([1.75], np.float32)
                sb = dev.create_gpu_buffer_with_data(sval)
                capi.call("agpu_scalar", h, op, dt, vp(cache[key]), vp(sb), vp(dout), n)
            got = dev.retrive_data(dout, 4 * n + 64, pipeline=p)
            if O is not None:
                exp = O.unary(op, dt, src) if kind == "unary" else O.scalar(op, dt, src, sval)
                g = got[: 4 * n].view(npd)
                if npd == np.float32:
                    # transcendental kernels are pinned to 1 ULP elsewhere (test_gpu_parity); here: exact functions bit for
                    # bit, the others within 1 ULP — a clobbered store is off by far more than that
                    gi, ei = g.view(np.int32).astype(np.int64), np.asarray(exp, np.float32).view(np.int32).astype(np.int64)
                    nan = np.isnan(g) | np.isnan(exp)
                    if not np.array_equal(np.isnan(g), np.isnan(exp)) or (np.abs(gi - ei)[~nan] > 1).any():
                        bad.append(name)
                elif not np.array_equal(g, np.asarray(exp).view(npd)):
                    bad.append(name)
        if not (got[4 * n:] == 0xA5).all():
            bad.append(name + " (wrote past the end)")
        out[name] = hashlib.sha256(got.tobytes()).hexdigest()[:16]
    return out, bad


if __name__ == "__main__":
    res, bad = main(check_oracle="--oracle" in sys.argv)
    sys.stdout.write(json.dumps({"lib": capi.LIB_PATH, "sites": res, "bad": bad}) + "\n")
