"""GPU: the tiles-per-block tuning ("tiles"; heavy_tiles / cast_tiles / table_tiles until round 6: a block walks k tiles a grid apart and issues the next tile's loads
before it evaluates the current one — elementwise.hip tile_run) never change a result: every kernel that has the walk, at k = 1 … 5 and with a
forced small grid, against its own k = default output bit for bit and against the oracle, at sizes around the tile boundaries."""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from arrow_gpu_amd import _capi as capi

pytestmark = pytest.mark.gpu

SIZES = [1, 1023, 1024, 1025, 4096 * 3 + 5, 65536 * 2 + 17, 1_000_003, 3_333_337]


class _Step(C.Structure):
    _fields_ = [("op", C.c_int32), ("kind", C.c_int32), ("operand", C.c_void_p)]


@pytest.fixture(scope="module")
def ctx():
    from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice

    dev = GpuDevice(0)
    return dev, ArrowComputePipeline(dev, "tiles")


def vp(b):
    return C.c_void_p(b.ptr)


@pytest.mark.parametrize("n", SIZES)
def test_every_walk_kernel_is_invariant_under_tiles_per_block(ctx, n):
    dev, p = ctx
    h = p._handle
    rng = np.random.default_rng(n)
    f = (rng.standard_normal(n) * 10.0 ** rng.integers(-2, 3, n)).astype(np.float32)
    f[: min(n, 8)] = [0.0, -0.0, np.inf, -np.inf, np.nan, 1e7, -1e7, 1e-40][: min(n, 8)]
    fpos = np.abs(f) + np.float32(1e-3)
    f2 = rng.uniform(-3, 3, n).astype(np.float32)
    u8 = rng.integers(0, 256, n, dtype=np.uint8)
    i16 = rng.integers(-32768, 32768, n, dtype=np.int64).astype(np.int16)
    df, dpos, df2, du8, di16 = (dev.create_gpu_buffer_with_data(x) for x in (f, fpos, f2, u8, i16))
    sc = dev.create_gpu_buffer_with_data(np.array([0.37], np.float32))
    out = dev.create_empty_buffer(4 * n + 16)
    st = (_Step * 2)()
    st[0].op, st[0].kind, st[0].operand = capi.OP_MUL, 1, sc.ptr
    st[1].op, st[1].kind, st[1].operand = capi.UN_SIN, 0, None
    kernels = {
        "heavy": [("sin_f32", lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.F32, vp(df), vp(out), n), lambda: O.unary(O.UN_SIN, O.F32, f), 1),
                        ("cos_f32", lambda: capi.call("agpu_unary", h, capi.UN_COS, capi.F32, vp(df), vp(out), n), lambda: O.unary(O.UN_COS, O.F32, f), 1),
                        ("sinh_f32", lambda: capi.call("agpu_unary", h, capi.UN_SINH, capi.F32, vp(df2), vp(out), n), lambda: O.unary(O.UN_SINH, O.F32, f2), 1)],
        "table": [("log_f32", lambda: capi.call("agpu_unary", h, capi.UN_LOG, capi.F32, vp(dpos), vp(out), n), lambda: O.unary(O.UN_LOG, O.F32, fpos), 1),
                        ("pow_f32", lambda: capi.call("agpu_binary", h, capi.OP_POW, capi.F32, vp(dpos), vp(df2), vp(out), n), lambda: O.binary(O.OP_POW, O.F32, fpos, f2), 1),
                        ("pow_f32_scalar", lambda: capi.call("agpu_scalar", h, capi.OP_POW, capi.F32, vp(dpos), vp(sc), vp(out), n),
                         lambda: O.scalar(O.OP_POW, O.F32, fpos, np.array([0.37], np.float32)), 1),
                        ("sin_u8", lambda: capi.call("agpu_unary", h, capi.UN_SIN, capi.U8, vp(du8), vp(out), n), lambda: O.unary(O.UN_SIN, O.U8, u8), 1),
                        ("cos_i16", lambda: capi.call("agpu_unary", h, capi.UN_COS, capi.I16, vp(di16), vp(out), n), lambda: O.unary(O.UN_COS, O.I16, i16), 1)],
        "cast": [("cast_u8_f32", lambda: capi.call("agpu_cast", h, capi.U8, capi.F32, vp(du8), vp(out), n), lambda: u8.astype(np.float32), 0),
                       ("cast_i16_f32", lambda: capi.call("agpu_cast", h, capi.I16, capi.F32, vp(di16), vp(out), n), lambda: i16.astype(np.float32), 0),
                       ("cast_i16_chain", lambda: capi.call("agpu_fused_cast_chain", h, capi.I16, vp(di16), C.cast(st, C.c_void_p), 2, vp(out), n),
                        lambda: O.unary(O.UN_SIN, O.F32, O.scalar(O.OP_MUL, O.F32, i16.astype(np.float32), np.array([0.37], np.float32))), 1),
                       ("cast_u8_chain", lambda: capi.call("agpu_fused_cast_chain", h, capi.U8, vp(du8), C.cast(st, C.c_void_p), 2, vp(out), n),
                        lambda: O.unary(O.UN_SIN, O.F32, O.scalar(O.OP_MUL, O.F32, u8.astype(np.float32), np.array([0.37], np.float32))), 1)],
    }
    try:
        key = "tiles"
        for family, rows in kernels.items():
            for name, launch, oracle, ulp in rows:
                p.set_tuning(key, 0)
                capi.call("agpu_memset", h, vp(out), 0xEE, 4 * n + 16)
                launch()
                ref = dev.retrive_data(out, 4 * n + 16, pipeline=p).copy()
                assert (ref[4 * n:] == 0xEE).all(), (name, "wrote behind the column")
                got = ref[: 4 * n].view(np.float32)
                exp = np.asarray(oracle(), np.float32)
                if ulp == 0:
                    assert got.tobytes() == exp.tobytes(), name
                else:  # ≤ 1 ULP against the oracle (f64 libm rounded once); NaN ↔ NaN, ±inf exact
                    fin = np.isfinite(exp) & np.isfinite(got)
                    assert np.array_equal(np.isnan(exp), np.isnan(got)) and np.array_equal(got[np.isinf(exp)], exp[np.isinf(exp)]), name
                    d = np.abs(got[fin].view(np.int32).astype(np.int64) - exp[fin].view(np.int32).astype(np.int64))
                    assert d.size == 0 or d.max() <= 1, (name, int(d.max()))
                for k, grid, lds in ((1, 0, 0), (2, 0, -1), (3, 0, 20000), (5, 0, 0), (0, 7, 4096), (4, 3, 0), (0, 0, -1), (0, 0, 70000)):
                    p.set_tuning(key, k)
                    p.set_tuning("stream_grid", grid)
                    p.set_tuning("wave_lds", lds)  # the occupancy cap (round 5): any value, same results
                    capi.call("agpu_memset", h, vp(out), 0xEE, 4 * n + 16)
                    launch()
                    again = dev.retrive_data(out, 4 * n + 16, pipeline=p)
                    assert again.tobytes() == ref.tobytes(), (name, key, k, grid)
                p.set_tuning("stream_grid", 0)
                p.set_tuning("wave_lds", 0)
                p.set_tuning(key, 0)
    finally:
        for key in ("tiles", "stream_grid", "wave_lds"):
            p.set_tuning(key, 0)
