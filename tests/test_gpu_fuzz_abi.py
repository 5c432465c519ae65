"""GPU: fuzzing at the C-ABI level — random kernel family × dtype × length × pointer mis-alignment against the CPU
oracle.  Lengths are drawn around every tile boundary of the kernels (64 / 256 / 1024 / 4096 / 65536 rows) and pointers
are offset by whole elements, so every launch mixes the vector path, its tails and the element-granular fallbacks.
Everything is bit-exact (transcendentals are excluded here; they have their own ULP sweeps)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle as O
from arrow_gpu_amd import _capi as capi
from gpu_util import ALL_DTYPES, INT_DTYPES, NP, Dev, bits_equal, nan_aware_bits_equal, rand_values

pytestmark = pytest.mark.gpu

EDGES = [64, 256, 1024, 2048, 4096, 8192, 65536]
F32_BIN = [capi.OP_ADD, capi.OP_SUB, capi.OP_MUL, capi.OP_DIV, capi.OP_REM, capi.OP_MIN, capi.OP_MAX]
INT32_BIN = F32_BIN + [capi.OP_AND, capi.OP_OR, capi.OP_XOR]
SMALL_BIN = [capi.OP_ADD, capi.OP_SUB, capi.OP_MUL, capi.OP_MIN, capi.OP_MAX, capi.OP_AND, capi.OP_OR, capi.OP_XOR]
CASTS = [(capi.I8, t) for t in (capi.U8, capi.U16, capi.U32, capi.I16, capi.I32, capi.F32)] + \
        [(capi.I16, t) for t in (capi.I32, capi.U16, capi.U32, capi.F32)] + \
        [(capi.U8, t) for t in (capi.U16, capi.U32, capi.I8, capi.I16, capi.I32, capi.F32)] + \
        [(capi.U16, t) for t in (capi.U32, capi.I16, capi.I32, capi.F32)] + [(capi.F32, capi.U8)]


TUNING_KEYS = ("stream_grid", "cmp_variant", "gather_bucket", "h2d_mode", "tiles", "wave_lds", "sync_spin")  # all of them


@pytest.fixture(scope="module")
def _dev():
    return Dev()


@pytest.fixture()
def D(_dev):
    yield _dev
    _dev.release()
    for key in TUNING_KEYS:
        _dev.p.set_tuning(key, 0)


def pick_n(rng):
    r = rng.random()
    if r < 0.08:
        return int(rng.integers(0, 4))
    if r < 0.6:
        return max(0, int(EDGES[rng.integers(len(EDGES))] * rng.integers(1, 4) + rng.integers(-3, 4)))
    return int(rng.integers(1, 200_000))


def off(rng, width):
    """Pointer offset in bytes: usually 0 (16-byte aligned), sometimes 1–3 elements."""
    return 0 if rng.random() < 0.6 else int(rng.integers(1, 4)) * width


def bin_ops(dtype):
    if dtype == capi.F32:
        return F32_BIN
    return INT32_BIN if NP[dtype]().itemsize == 4 else SMALL_BIN


@pytest.mark.parametrize("seed", range(int(os.environ.get("AGPU_FUZZ_BASE", "0")), int(os.environ.get("AGPU_FUZZ_BASE", "0")) + int(os.environ.get("AGPU_FUZZ_SEEDS", "40"))))  # soaks: AGPU_FUZZ_SEEDS = how many, AGPU_FUZZ_BASE = the first
def test_random_abi_calls_match_the_oracle(D, seed):
    rng = np.random.default_rng(5000 + seed)
    for it in range(12):
        fam = ("binary", "scalar", "unary", "compare", "compare_validity", "cast", "bitmap", "reduce", "take", "merge",
               "chain", "shift", "put", "take_bits", "take_validity", "put_bits", "cast_chain", "stats", "take_columns")[rng.integers(19)]
        # EVERY tuning key the ABI has (include/arrow_gpu.h: seven since round 6) is drawn in every iteration: results never depend on them.
        # tiles per block of the prefetching kernels (heavy unary kernels, casts, cast-headed chains, table kernels): any value, same results
        D.p.set_tuning("tiles", int(rng.integers(0, 9)) if rng.random() < 0.5 else 0)
        D.p.set_tuning("stream_grid", int((0, 0, 0, 7, 1024)[rng.integers(5)]))  # persistent grids: the kernels' grid-stride loops
        D.p.set_tuning("cmp_variant", int(rng.integers(0, 2)))                    # ballot compare / vector compare
        D.p.set_tuning("h2d_mode", int(rng.integers(0, 4)))                       # (staging route of Arrow imports: no call of the fuzz reads it)
        D.p.set_tuning("wave_lds", int((0, 0, -1, 3000, 40000)[rng.integers(5)]))  # occupancy cap of sin / cos, the widening casts, the 8-bit table kernels
        D.p.set_tuning("sync_spin", int((0, 0, -1, 1, 30)[rng.integers(5)]))  # the mailbox waits (R5.10): on, off, a spin budget that mostly / sometimes runs out
        # take / put: the direct kernels (auto at these sizes; 1), the forced pipelines (2: merge-back take, pair-pipeline put) — small, ragged,
        # mis-aligned inputs through every form
        D.p.set_tuning("gather_bucket", int((0, 1, 2, 4, 4)[rng.integers(5)]))  # 4: pipelines + the device-side probe at any size
        n = pick_n(rng)
        dtype = ALL_DTYPES[rng.integers(len(ALL_DTYPES))]
        w = NP[dtype]().itemsize
        what = f"seed {seed} it {it}: {fam} dtype {dtype} n {n}"
        a, b = rand_values(dtype, n, seed * 100 + it), rand_values(dtype, n, seed * 100 + it + 50)
        if fam in ("binary", "scalar"):
            op = bin_ops(dtype)[rng.integers(len(bin_ops(dtype)))]
            out = D.empty(max(n * w, 1), offset_bytes=off(rng, w))
            if fam == "binary":
                D.call("agpu_binary", op, dtype, D.up(a, off(rng, w)).vp, D.up(b, off(rng, w)).vp, out.vp, n)
                exp = O.binary(op, dtype, a, b)
            else:
                s = rand_values(dtype, 1, seed + it, special=False)
                D.call("agpu_scalar", op, dtype, D.up(a, off(rng, w)).vp, D.up(s).vp, out.vp, n)
                exp = O.scalar(op, dtype, a, s)
            assert nan_aware_bits_equal(D.down(out, NP[dtype], n), exp), what
        elif fam == "unary":
            ops = [capi.UN_NEG, capi.UN_ABS] + ([capi.UN_SQRT] if dtype == capi.F32 else [capi.UN_NOT])
            op = ops[rng.integers(len(ops))]
            out = D.empty(max(n * w, 1), offset_bytes=off(rng, w))
            D.call("agpu_unary", op, dtype, D.up(a, off(rng, w)).vp, out.vp, n)
            assert nan_aware_bits_equal(D.down(out, NP[dtype], n), O.unary(op, dtype, a)), what
        elif fam in ("compare", "compare_validity"):
            op = int(rng.integers(0, 5))
            out = D.empty(O.bitmap_bytes(n) + 8)
            nb = O.bitmap_bytes(n)
            if fam == "compare":
                D.call("agpu_compare", op, dtype, D.up(a, off(rng, w)).vp, D.up(b, off(rng, w)).vp, out.vp, n)
            else:
                va, vb = O.synth_bits(n, seed, it, 0.8), O.synth_bits(n, seed + 1, it, 0.8)
                outv = D.empty(nb + 8)
                D.call("agpu_compare_validity", op, dtype, D.up(a, off(rng, w)).vp, D.up(b, off(rng, w)).vp, D.up(va).vp,
                       D.up(vb).vp, out.vp, outv.vp, n)
                assert bits_equal(D.down(outv, np.uint8, nb), O.bitmap_binary(O.OP_AND, va, vb, n)), what
            assert bits_equal(D.down(out, np.uint8, nb), O.compare(op, dtype, a, b)), what
        elif fam == "cast":
            frm, to = CASTS[rng.integers(len(CASTS))]
            x = rand_values(frm, n, seed * 7 + it)
            wi, wo = NP[frm]().itemsize, NP[to]().itemsize
            out = D.empty(max(n * wo, 1), offset_bytes=off(rng, wo))
            D.call("agpu_cast", frm, to, D.up(x, off(rng, wi)).vp, out.vp, n)
            assert bits_equal(D.down(out, NP[to], n), O.cast(frm, to, x)), what
        elif fam == "bitmap":
            va, vb = O.synth_bits(n, seed, it, 0.5), O.synth_bits(n, seed + 9, it, 0.5)
            nb = O.bitmap_bytes(n)
            out = D.empty(nb + 8)
            op = (capi.OP_AND, capi.OP_OR, capi.OP_XOR)[rng.integers(3)]
            D.call("agpu_bitmap_binary", op, D.up(va).vp, D.up(vb).vp, out.vp, n)
            assert bits_equal(D.down(out, np.uint8, nb), O.bitmap_binary(op, va, vb, n)), what
            cnt = D.empty(16)
            D.call("agpu_bitmap_popcount", D.up(va).vp, n, cnt.vp)
            assert int(D.down(cnt, np.uint64, 1)[0]) == int(np.unpackbits(va, bitorder="little")[:n].sum()), what
        elif fam == "reduce":
            rt = (capi.F32, capi.I32, capi.U32)[rng.integers(3)]
            x = rand_values(rt, n, seed * 3 + it, special=False)
            op = (capi.RED_SUM, capi.RED_MIN, capi.RED_MAX)[rng.integers(3)]
            v = O.synth_bits(n, seed, it + 3, 0.7) if rng.random() < 0.5 else None
            if n == 0 or (v is not None and not np.unpackbits(v, bitorder="little")[:n].any()):
                continue
            out = D.empty(16)
            D.call("agpu_reduce", op, rt, D.up(x, off(rng, 4)).vp, D.up(v).vp if v is not None else None, n, out.vp)
            assert nan_aware_bits_equal(D.down(out, NP[rt], 1), np.atleast_1d(O.reduce(op, rt, x, v)).astype(NP[rt])), what
        elif fam == "stats":  # the one-pass statistics against the four reductions' oracle values (small sizes: the fallback; the fused form: test_gpu_stats.py)
            x = rand_values(capi.F32, n, seed * 3 + it, special=False)
            v = O.synth_bits(n, seed, it + 3, 0.7) if rng.random() < 0.5 else None
            if n == 0 or (v is not None and not np.unpackbits(v, bitorder="little")[:n].any()):
                continue
            rec = D.empty(32)
            D.call("agpu_reduce_stats_f32", D.up(x, off(rng, 4)).vp, D.up(v).vp if v is not None else None, n, rec.vp)
            raw = D.down(rec, np.uint8, 24)
            f3 = raw[:12].view(np.float32)
            for j, op in enumerate((capi.RED_SUM, capi.RED_MIN, capi.RED_MAX)):
                assert nan_aware_bits_equal(f3[j:j + 1], np.atleast_1d(O.reduce(op, capi.F32, x, v)).astype(np.float32)), (what, op)
            assert int(raw[12:16].view(np.uint32)[0]) == 0, what
        elif fam == "take_columns":
            if n == 0:
                continue
            k = pick_n(rng)
            idx = rng.integers(0, n, k).astype(np.uint32)
            if rng.random() < 0.4:
                idx = np.sort(idx)
            ncol = int(rng.integers(1, 5))
            cdt = [(capi.U32, capi.U16, capi.U8, capi.I32, capi.F32)[rng.integers(5)] for _ in range(ncol)]
            cols = [rand_values(t, n, seed * 23 + it + j, special=False) for j, t in enumerate(cdt)]
            vbs = [O.synth_bits(n, seed, it + 9 + j, 0.8) if rng.random() < 0.4 else None for j in range(ncol)]
            dcols = [D.up(c) for c in cols]
            dvb = [D.up(vb) if vb is not None else None for vb in vbs]
            outs = [D.empty(max(k * c.dtype.itemsize, 1)) for c in cols]
            outv = [D.empty(O.bitmap_bytes(k) + 8) if vb is not None else None for vb in vbs]
            widths = (C.c_int32 * ncol)(*[c.dtype.itemsize for c in cols])
            vals = (C.c_void_p * ncol)(*[d.vp.value for d in dcols])
            vbp = (C.c_void_p * ncol)(*[d.vp.value if d is not None else None for d in dvb])
            outp = (C.c_void_p * ncol)(*[o.vp.value for o in outs])
            ovp = (C.c_void_p * ncol)(*[o.vp.value if o is not None else None for o in outv])
            D.call("agpu_take_columns_validity", ncol, widths, vals, vbp, n, D.up(idx).vp, outp, ovp, k)
            for c, o, vb, ov in zip(cols, outs, vbs, outv):
                assert bits_equal(D.down(o, c.dtype, k), O.take(c.dtype.itemsize, c, idx)), what
                if vb is not None and k:
                    assert bits_equal(D.down(ov, np.uint8, O.bitmap_bytes(k)), O.take_bits(vb, n, idx)), what
            assert D.status("agpu_pipeline_sync") == capi.OK
        elif fam == "take":
            if n == 0:
                continue
            k = pick_n(rng)
            idx = rng.integers(0, n, k).astype(np.uint32)
            if rng.random() < 0.4:
                idx = np.sort(idx)  # local indices: the probe's other outcome
            out = D.empty(max(k * w, 1), offset_bytes=off(rng, w))
            D.call("agpu_take", w, D.up(a, off(rng, w)).vp, n, D.up(idx, off(rng, 4)).vp, out.vp, k)
            assert bits_equal(D.down(out, NP[dtype], k), O.take(w, a, idx)), what
            assert D.status("agpu_pipeline_sync") == capi.OK
        elif fam == "merge":
            m = O.synth_bits(n, seed, it + 5, 0.5)
            out = D.empty(max(n * w, 1), offset_bytes=off(rng, w) if w == 4 else 0)
            D.call("agpu_merge", w, D.up(a, off(rng, w) if w == 4 else 0).vp, D.up(b).vp, D.up(m).vp, out.vp, n)
            assert bits_equal(D.down(out, NP[dtype], n), O.merge(w, a, b, m)), what
        elif fam == "shift":
            st = INT_DTYPES[rng.integers(len(INT_DTYPES))]
            ws = NP[st]().itemsize
            x = rand_values(st, n, seed * 5 + it)
            amounts = rng.integers(0, 40, n).astype(np.uint32)  # includes amounts ≥ the type width and ≥ 32
            op = (capi.OP_SHL, capi.OP_SHR)[rng.integers(2)]
            out = D.empty(max(n * ws, 1), offset_bytes=off(rng, ws))
            D.call("agpu_binary", op, st, D.up(x, off(rng, ws)).vp, D.up(amounts, off(rng, 4)).vp, out.vp, n)
            assert bits_equal(D.down(out, NP[st], n), O.binary(op, st, x, amounts)), what
        elif fam == "put":
            if n == 0:
                continue
            n_dst = n + int(rng.integers(0, 100))
            k = min(pick_n(rng), n_dst)
            dst = rand_values(dtype, n_dst, seed * 19 + it)
            si = rng.integers(0, n, k).astype(np.uint32)
            di = rng.permutation(n_dst)[:k].astype(np.uint32)  # unique destinations
            r = rng.random()  # local columns: both, the source only, the destination only — the put's four device-side forms
            if r < 0.2:
                si, di = np.sort(si), np.sort(di)
            elif r < 0.4:
                si = np.sort(si)
            elif r < 0.6:
                di = np.sort(di)
            ddst = D.up(dst, off(rng, w))
            D.call("agpu_put_bounded", w, D.up(a, off(rng, w)).vp, n, D.up(si, off(rng, 4)).vp, ddst.vp, n_dst,
                   D.up(di, off(rng, 4)).vp, k)
            assert bits_equal(D.down(ddst, NP[dtype], n_dst), O.put(w, a, si, dst, di)), what
            assert D.status("agpu_pipeline_sync") == capi.OK
        elif fam == "put_bits":
            if n == 0:
                continue
            n_dst = n + int(rng.integers(0, 100))
            k = min(pick_n(rng), n_dst)
            src_b, dst_b = O.synth_bits(n, seed, it + 11, 0.5), O.synth_bits(n_dst, seed, it + 12, 0.5)
            si = rng.integers(0, n, k).astype(np.uint32)
            di = rng.permutation(n_dst)[:k].astype(np.uint32)  # unique destinations
            if rng.random() < 0.4:
                si, di = np.sort(si), np.sort(di)
            ddst = D.up(dst_b)
            D.call("agpu_put_bits_bounded", D.up(src_b).vp, n, D.up(si, off(rng, 4)).vp, ddst.vp, n_dst, D.up(di, off(rng, 4)).vp, k)
            assert bits_equal(D.down(ddst, np.uint8, O.bitmap_bytes(n_dst)), O.put_bits(src_b, si, dst_b, di)), what
            assert D.status("agpu_pipeline_sync") == capi.OK
        elif fam == "take_validity":
            if n == 0:
                continue
            k = pick_n(rng)
            idx = rng.integers(0, n, k).astype(np.uint32)
            if rng.random() < 0.4:
                idx = np.sort(idx)
            vb = O.synth_bits(n, seed, it + 9, 0.6)
            out = D.empty(max(k * w, 1), offset_bytes=off(rng, w))
            outv = D.empty(O.bitmap_bytes(k) + 8)
            D.call("agpu_take_validity", w, D.up(a, off(rng, w)).vp, n, D.up(vb).vp, D.up(idx, off(rng, 4)).vp, out.vp, outv.vp, k)
            assert bits_equal(D.down(out, NP[dtype], k), O.take(w, a, idx)), what
            assert bits_equal(D.down(outv, np.uint8, O.bitmap_bytes(k)), O.take_bits(vb, n, idx)), what
            assert D.status("agpu_pipeline_sync") == capi.OK
        elif fam == "take_bits":
            if n == 0:
                continue
            bits_in = O.synth_bits(n, seed, it + 7, 0.5)
            k = pick_n(rng)
            idx = rng.integers(0, n, k).astype(np.uint32)
            if rng.random() < 0.4:
                idx = np.sort(idx)
            outb = D.empty(O.bitmap_bytes(k) + 8)
            D.call("agpu_take_bits", D.up(bits_in).vp, n, D.up(idx, off(rng, 4)).vp, outb.vp, k)
            assert bits_equal(D.down(outb, np.uint8, O.bitmap_bytes(k)), O.take_bits(bits_in, n, idx)), what
        elif fam == "cast_chain":  # a widening cast at the head of 0–4 exact f32 steps, one launch (8-bit sources of ≥ 65 536 rows: the table route)

            frm = (capi.U8, capi.I8, capi.U16, capi.I16)[rng.integers(4)]
            wi = NP[frm]().itemsize
            x = rand_values(frm, n, seed * 23 + it)
            k = int(rng.integers(0, 5))

            class CStep(C.Structure):
                _fields_ = [("op", C.c_int32), ("kind", C.c_int32), ("operand", C.c_void_p)]

            steps = (CStep * max(k, 1))()
            exp = O.cast(frm, capi.F32, x)
            for s in range(k):
                kind = int(rng.integers(0, 3))
                if kind == 0:
                    op = (capi.UN_NEG, capi.UN_ABS, capi.UN_SQRT)[rng.integers(3)]
                    steps[s].op, steps[s].kind, steps[s].operand = op, 0, None
                    exp = O.unary(op, capi.F32, exp)
                else:
                    ops = [capi.OP_ADD, capi.OP_SUB, capi.OP_MUL, capi.OP_DIV, capi.OP_MIN, capi.OP_MAX]
                    op = ops[rng.integers(len(ops))]
                    y = rand_values(capi.F32, n if kind == 2 else 1, seed * 29 + it * 7 + s, special=kind == 2)
                    steps[s].op, steps[s].kind, steps[s].operand = op, kind, D.up(y, off(rng, 4) if kind == 2 else 0).vp.value
                    exp = O.binary(op, capi.F32, exp, y) if kind == 2 else O.scalar(op, capi.F32, exp, y)
            out = D.empty(max(4 * n, 1), offset_bytes=off(rng, 4))
            D.call("agpu_fused_cast_chain", frm, D.up(x, off(rng, wi)).vp, C.cast(steps, C.c_void_p), k, out.vp, n)
            assert nan_aware_bits_equal(D.down(out, np.float32, n), exp), what
        else:  # chain of 2–5 exact steps on a 32-bit column, with or without a terminal compare

            ct = (capi.F32, capi.I32, capi.U32)[rng.integers(3)]
            x = rand_values(ct, n, seed * 11 + it)
            k = int(rng.integers(1, 6))

            class Step(C.Structure):
                _fields_ = [("op", C.c_int32), ("kind", C.c_int32), ("operand", C.c_void_p)]

            steps = (Step * k)()
            exp = x
            for s in range(k):
                kind = int(rng.integers(0, 3))
                if kind == 0:
                    op = (capi.UN_NEG, capi.UN_ABS)[rng.integers(2)]
                    steps[s].op, steps[s].kind, steps[s].operand = op, 0, None
                    exp = O.unary(op, ct, exp)
                else:
                    ops = [capi.OP_ADD, capi.OP_SUB, capi.OP_MUL, capi.OP_MIN, capi.OP_MAX]
                    op = ops[rng.integers(len(ops))]
                    y = rand_values(ct, n if kind == 2 else 1, seed * 13 + it * 7 + s, special=kind == 2)
                    steps[s].op, steps[s].kind, steps[s].operand = op, kind, D.up(y, off(rng, 4) if kind == 2 else 0).vp.value
                    exp = O.binary(op, ct, exp, y) if kind == 2 else O.scalar(op, ct, exp, y)
            if rng.random() < 0.5:
                out = D.empty(max(4 * n, 1), offset_bytes=off(rng, 4))
                D.call("agpu_fused_chain", ct, D.up(x, off(rng, 4)).vp, C.cast(steps, C.c_void_p), k, out.vp, n)
                assert nan_aware_bits_equal(D.down(out, NP[ct], n), exp), what
            else:
                c = rand_values(ct, n, seed * 17 + it)
                cmp_op = int(rng.integers(0, 5))
                outb = D.empty(O.bitmap_bytes(n) + 8)
                D.call("agpu_fused_chain_compare", ct, D.up(x, off(rng, 4)).vp, C.cast(steps, C.c_void_p), k, cmp_op, 2,
                       D.up(c, off(rng, 4)).vp, outb.vp, n)
                assert bits_equal(D.down(outb, np.uint8, O.bitmap_bytes(n)), O.compare(cmp_op, ct, exp, c)), what
