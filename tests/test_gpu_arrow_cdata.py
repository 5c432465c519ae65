"""GPU: Arrow C Data Interface at the C ABI (agpu_import_arrow / agpu_export_arrow) driven by pyarrow's
`_export_to_c` / `_import_from_c`, the host↔HBM staging engine in its three modes, and the overlapped chunk pipeline.
Sliced arrays, nulls, every supported type, empty arrays, arrays large enough for the threaded staging path.
[ref: PrimitiveArrayGpu::from_slice / raw_values crates/array/src/array/primitive_array_gpu.rs:22-104 — the
reference's only ingest / egress]"""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from arrow_gpu_amd import _capi as capi

pa = pytest.importorskip("pyarrow")
pytestmark = pytest.mark.gpu

TYPES = [pa.float32(), pa.int32(), pa.uint32(), pa.int16(), pa.uint16(), pa.int8(), pa.uint8(), pa.date32(), pa.bool_()]


def make(typ, n, seed, null_p):
    rng = np.random.default_rng(seed)
    if typ == pa.bool_():
        vals = rng.random(n) < 0.5
    elif typ == pa.float32():
        vals = rng.standard_normal(n).astype(np.float32)
    elif typ == pa.date32():
        vals = rng.integers(-50000, 50000, n).astype(np.int32)
    else:
        info = np.iinfo(typ.to_pandas_dtype())
        vals = rng.integers(info.min, int(info.max) + 1, n, dtype=np.int64).astype(typ.to_pandas_dtype())
    mask = rng.random(n) < null_p if null_p else None
    return pa.array(vals, type=typ, mask=mask)


@pytest.mark.parametrize("typ", TYPES, ids=str)
@pytest.mark.parametrize("n,off,ln,null_p", [(0, 0, 0, 0.0), (1, 0, 1, 0.0), (100, 0, 100, 0.2), (100, 3, 64, 0.2), (1000, 8, 900, 0.3),
                                             (70_001, 13, 65_000, 0.1), (3_000_000, 5, 2_999_000, 0.05), (3_000_000, 0, 3_000_000, 0.0)])
def test_round_trip_through_the_c_data_interface(ag, typ, n, off, ln, null_p):
    dev = ag.GPU_DEVICE()
    arr = make(typ, n, n + off, null_p).slice(off, ln)
    g = ag.from_arrow(arr, dev)
    assert g.len == ln and (g.null_buffer is not None) == (arr.null_count > 0)
    back = ag.to_arrow(g)
    assert back.type == arr.type and len(back) == ln and back.null_count == arr.null_count
    assert back.equals(arr)
    if ln and g.null_buffer is not None:  # word-aligned bitmap, padding bits zero (the kernels rely on it)
        raw = dev.retrive_data(g.null_buffer.bit_buffer, O.bitmap_bytes(ln))
        assert O.bitmap_popcount(raw, len(raw) * 8) == ln - arr.null_count


def test_imported_columns_feed_kernels(ag):
    import pyarrow.compute as pc

    dev = ag.GPU_DEVICE()
    a = make(pa.float32(), 2_000_000, 1, 0.1).slice(7, 1_900_000)
    b = make(pa.float32(), 2_000_000, 2, 0.1).slice(11, 1_900_000)
    got = ag.to_arrow(ag.from_arrow(a, dev).add(ag.from_arrow(b, dev)))
    assert got.equals(pc.add(a, b))
    x = make(pa.int32(), 1_000_000, 3, 0.2).slice(3, 999_000)
    y = make(pa.int32(), 1_000_000, 4, 0.2).slice(3, 999_000)
    assert ag.to_arrow(ag.from_arrow(x, dev).eq(ag.from_arrow(y, dev))).equals(pc.equal(x, y))


def test_unsupported_arrow_layouts_are_rejected(ag):
    dev = ag.GPU_DEVICE()
    with pytest.raises(ag.ArrowErrorGPU):
        ag.from_arrow(pa.array(["a", "b"]), dev)
    with pytest.raises(ag.ArrowErrorGPU):
        ag.from_arrow(pa.array([1.0, 2.0], type=pa.float64()), dev)


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_staged_copy_modes_move_the_same_bytes(ag, mode):
    dev = ag.GPU_DEVICE()
    p = ag.ArrowComputePipeline(dev, "staging")
    p.set_tuning("h2d_mode", mode)
    for nbytes in (1, 4095, (4 << 20) + 3, (37 << 20) + 12345):
        src = np.random.default_rng(nbytes).integers(0, 256, nbytes, dtype=np.uint8)
        buf = dev.create_empty_buffer(nbytes + 64)
        capi.call("agpu_memset", p._handle, C.c_void_p(buf.ptr), 0xEE, nbytes + 64)
        capi.call("agpu_staged_copy", p._handle, C.c_void_p(buf.ptr), C.c_void_p(src.ctypes.data), nbytes, 1)
        dst = np.zeros(nbytes + 64, np.uint8)
        capi.call("agpu_staged_copy", p._handle, C.c_void_p(buf.ptr), C.c_void_p(dst.ctypes.data), nbytes + 64, 0)
        assert np.array_equal(dst[:nbytes], src) and (dst[nbytes:] == 0xEE).all(), (mode, nbytes)


def test_map_chunks_overlapped_pipeline(ag):
    dev = ag.GPU_DEVICE()
    n = 40_000_003
    rng = np.random.default_rng(0)
    a, b = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    out = np.empty(n, np.float32)

    def launch(p, ins, o, rows):
        capi.call("agpu_binary", p._handle, capi.OP_ADD, capi.F32, C.c_void_p(ins[0].ptr), C.c_void_p(ins[1].ptr), C.c_void_p(o.ptr), rows)

    info = ag.interop.map_chunks(dev, [a, b], out, 1 << 22, launch)
    assert np.array_equal(out, a + b)
    print(f"map_chunks f32 add from and to pageable host memory: {info['GBps_host_bytes']:.1f} GB/s of host bytes, {info['chunks']} chunks")


def test_map_chunks_ends_its_uploader_when_the_consumer_fails(ag):
    """launch() raising mid-way must not leave the uploader thread parked on its semaphore (the interpreter could
    not exit)"""
    import threading

    dev = ag.GPU_DEVICE()
    n = 8_000_000
    a = np.ones(n, np.float32)
    out = np.empty(n, np.float32)
    calls = []

    def launch(p, ins, o, rows):
        calls.append(rows)
        if len(calls) == 2:
            raise RuntimeError("consumer failed")
        capi.call("agpu_copy", p._handle, C.c_void_p(o.ptr), C.c_void_p(ins[0].ptr), 4 * rows)

    before = threading.active_count()
    with pytest.raises(RuntimeError, match="consumer failed"):
        ag.interop.map_chunks(dev, [a], out, 1 << 20, launch)
    assert threading.active_count() == before
    info = ag.interop.map_chunks(dev, [a], out, 1 << 20, lambda p, ins, o, rows: capi.call(
        "agpu_copy", p._handle, C.c_void_p(o.ptr), C.c_void_p(ins[0].ptr), 4 * rows))
    assert info["chunks"] == 8 and np.array_equal(out, a)


def test_record_batch_imports_into_one_table_block(ag):
    """from_arrow_batch: every column's buffers out of ONE device block (agpu_import_arrow_table), values bit for bit,
    sliced inputs and nulls included; the columns stay ordinary arrays (kernels, to_arrow, independent lifetime)."""
    import pyarrow.compute as pc

    dev = ag.GPU_DEVICE()
    n = 300_007
    cols = {"f": make(pa.float32(), n + 5, 1, 0.1).slice(5), "g": make(pa.float32(), n + 5, 2, 0.0).slice(5),
            "i": make(pa.int32(), n + 13, 3, 0.2).slice(13), "b": make(pa.bool_(), n + 3, 4, 0.3).slice(3),
            "u": make(pa.uint8(), n, 5, 0.0), "d": make(pa.date32(), n, 6, 0.05)}
    got = ag.from_arrow_batch(pa.record_batch(list(cols.values()), names=list(cols)), dev)
    assert list(got) == list(cols)
    ptrs = []
    for name, arr in cols.items():
        g = got[name]
        assert g.len == n and (g.null_buffer is not None) == (arr.null_count > 0)
        assert ag.to_arrow(g).equals(arr), name
        ptrs.append(g.data.ptr)
        if g.null_buffer is not None:
            ptrs.append(g.null_buffer.bit_buffer.ptr)
    assert max(ptrs) - min(ptrs) < 64 << 20  # one block: 2 MiB strides + colours for these sizes
    assert len({q % (1 << 21) for q in ptrs[:4]}) > 1  # coloured starts, not all on 2 MiB boundaries
    s = got["f"].add(got["g"])
    assert ag.to_arrow(s).equals(pc.add(cols["f"], cols["g"]))
    del got["f"], got["b"]  # columns die independently; the rest stays valid
    assert ag.to_arrow(got["i"]).equals(cols["i"])


def test_record_batch_reader_through_the_c_stream_interface(ag):
    """Arrow C STREAM interface: the library pulls batch after batch from a pyarrow RecordBatchReader"""
    dev = ag.GPU_DEVICE()
    n = 250_003
    t = pa.table({"a": make(pa.float32(), n, 1, 0.1), "s": pa.array(["q"] * n), "b": make(pa.int16(), n, 2, 0.0),
                  "c": make(pa.bool_(), n, 3, 0.25)})
    batches = t.to_batches(max_chunksize=60_000)
    reader = pa.RecordBatchReader.from_batches(t.schema, batches)
    seen = 0
    for bi, cols in enumerate(ag.from_arrow_reader(reader, dev)):
        assert list(cols) == ["a", "b", "c"]  # the utf8 column has no GPU array type and is left out
        for name, g in cols.items():
            assert ag.to_arrow(g).equals(batches[bi].column(t.schema.names.index(name))), (bi, name)
        seen += 1
    assert seen == len(batches)
    reader = pa.RecordBatchReader.from_batches(t.schema, batches)
    only_b = list(ag.from_arrow_reader(reader, dev, columns=["b"]))
    assert len(only_b) == len(batches) and all(list(c) == ["b"] for c in only_b)
