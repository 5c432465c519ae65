"""GPU: Arrow IPC ↔ HBM through the C ABI (agpu_ipc_read_column / agpu_ipc_writer_write_device_batch):
what pyarrow serialised must arrive on the device bit for bit (all nine array types, nulls, several record batches,
stream / file / memory-mapped file), a kernel runs on it, and what the writer serialises from HBM must read back in
pyarrow equal to the oracle's result."""
import numpy as np
import pytest

import oracle as O
from arrow_gpu_amd import _capi as capi

pa = pytest.importorskip("pyarrow")
pytestmark = pytest.mark.gpu

from test_ipc_host import TYPES, make_table, serialise  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    from arrow_gpu_amd.gpu_utils import GpuDevice

    return GpuDevice(0)


@pytest.mark.parametrize("file_format", [False, True])
@pytest.mark.parametrize("n,batch_rows", [(1003, 257), (100_000, 100_000), (70_001, 16_384), (0, 5)])
def test_ipc_to_device_and_back(dev, file_format, n, batch_rows):
    from arrow_gpu_amd.ipc import IpcReader, write_ipc

    rng = np.random.default_rng(n + batch_rows + file_format)
    table = make_table(rng, n)
    data = serialise(table, file_format, batch_rows)
    batches = table.to_batches(max_chunksize=batch_rows)
    with IpcReader(data) as r:
        cols = r.read_all(dev)
    names = [f"col_{fmt}" for fmt, _, _ in TYPES]
    assert sorted(cols) == sorted(names) or n == 0
    for name in cols:
        for bi, arr in enumerate(cols[name]):
            exp = batches[bi].column(name)
            got = arr.to_arrow()
            assert got.type == exp.type and got.null_count == exp.null_count
            assert got.equals(exp), (name, bi)
    if n == 0:
        return
    # HBM → IPC: pyarrow must read exactly what went in
    out = write_ipc({k: cols[k] for k in names}, None, file_format)
    back = (pa.ipc.open_file if file_format else pa.ipc.open_stream)(pa.BufferReader(out)).read_all()
    back.validate(full=True)
    assert back.equals(table.select(names))


def test_kernel_between_ipc_in_and_ipc_out(dev, tmp_path):
    """file (mmap) → HBM → f32 add + i32 eq with nulls → file; checked against the oracle through pyarrow"""
    from arrow_gpu_amd.ipc import IpcReader, IpcWriter

    n = 300_007
    rng = np.random.default_rng(3)
    fa, fb = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    ia, ib = rng.integers(0, 4, n).astype(np.int32), rng.integers(0, 4, n).astype(np.int32)
    ma, mb = rng.random(n) < 0.1, rng.random(n) < 0.1
    t = pa.table({"fa": pa.array(fa, mask=ma), "fb": pa.array(fb, mask=mb), "ia": pa.array(ia, mask=mb), "ib": pa.array(ib, mask=ma)})
    src, dst = tmp_path / "in.arrow", tmp_path / "out.arrow"
    with pa.OSFile(str(src), "wb") as f, pa.ipc.new_file(f, t.schema) as w:
        for b in t.to_batches(max_chunksize=100_000):
            w.write_batch(b)
    from arrow_gpu_amd import BooleanArrayGPU, Float32ArrayGPU

    w = IpcWriter([("sum", Float32ArrayGPU, True), ("eq", BooleanArrayGPU, True)], str(dst), file_format=True)
    with IpcReader(str(src)) as r:
        assert r.num_batches == 4
        for b in range(r.num_batches):
            c = r.read_batch(b, dev)
            w.write_batch([c["fa"].add(c["fb"]), c["ia"].eq(c["ib"])])
    assert w.finish() is None
    got = pa.ipc.open_file(pa.memory_map(str(dst))).read_all()
    valid = ~(ma | mb)
    exp_sum = O.binary(O.OP_ADD, O.F32, fa, fb)
    g = got.column("sum").combine_chunks()
    assert np.array_equal(np.asarray(g.is_valid()), valid)
    assert np.array_equal(g.fill_null(0).to_numpy(zero_copy_only=False).view(np.uint32)[valid], exp_sum.view(np.uint32)[valid])
    e = got.column("eq").combine_chunks()
    assert np.array_equal(np.asarray(e.is_valid()), valid)
    assert np.array_equal(np.asarray(e.fill_null(False))[valid], (ia == ib)[valid])
    assert g.null_count == int((~valid).sum()) == e.null_count


def test_large_column_through_ipc(dev, tmp_path):
    """256 MiB f32 column + validity: file → HBM → file, checksum-equal, rates printed (page cache → HBM → page cache)"""
    import time

    from arrow_gpu_amd.ipc import IpcReader, write_ipc

    n = 1 << 26
    vals = O.synth_f32(n, 9, 0, -1, 1)
    bits = O.synth_bits(n, 10, 0, 0.9)
    arr = pa.Array.from_buffers(pa.float32(), n, [pa.py_buffer(bits.tobytes()), pa.py_buffer(vals)])
    path = tmp_path / "big.arrow"
    with pa.OSFile(str(path), "wb") as f, pa.ipc.new_file(f, pa.schema([("x", pa.float32())])) as w:
        w.write_batch(pa.record_batch([arr], names=["x"]))
    from arrow_gpu_amd.gpu_utils import ArrowComputePipeline

    rates = {}
    for mode in (1, 2, 1, 2):  # pageable hipMemcpy from the mapping vs threaded page-locked staging (4 KiB page-cache pages)
        p = ArrowComputePipeline(dev, "ipc.big")
        p.set_tuning("h2d_mode", mode)
        t0 = time.perf_counter()
        with IpcReader(str(path)) as r:
            x = r.read_column(0, 0, dev, p)
            p.sync()
        rates[mode] = vals.nbytes / (time.perf_counter() - t0) / 1e9
    assert x.len == n and x.null_buffer is not None
    out = tmp_path / "big_out.arrow"
    t1 = time.perf_counter()
    write_ipc({"x": x}, str(out), file_format=True)
    t2 = time.perf_counter()
    print(f"\nIPC file → HBM: h2d_mode 1 {rates[1]:.1f} GB/s, h2d_mode 2 {rates[2]:.1f} GB/s; HBM → IPC file {vals.nbytes / (t2 - t1) / 1e9:.1f} GB/s")
    back = pa.ipc.open_file(pa.memory_map(str(out))).read_all().column("x").chunk(0)
    assert back.null_count == arr.null_count
    assert back.equals(arr)


def test_lz4_compressed_file_to_device(dev, tmp_path):
    """An LZ4-compressed IPC file (what Feather V2 writes by default): decompressed by the library, uploaded, bit-identical"""
    from arrow_gpu_amd.ipc import IpcReader

    rng = np.random.default_rng(8)
    n = 400_000
    t = pa.table({"k": pa.array(np.repeat(np.arange(n // 50, dtype=np.int32), 50), mask=np.arange(n) % 13 == 0),
                  "x": pa.array(rng.standard_normal(n).astype(np.float32)), "s": pa.array(["ab"] * n),
                  "f": pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.2)})
    path = tmp_path / "c.arrow"
    try:
        opts = pa.ipc.IpcWriteOptions(compression="lz4")
    except Exception:
        pytest.skip("pyarrow built without lz4")
    with pa.OSFile(str(path), "wb") as f, pa.ipc.new_file(f, t.schema, options=opts) as w:
        for b in t.to_batches(max_chunksize=150_000):
            w.write_batch(b)
    with IpcReader(str(path)) as r:
        cols = r.read_all(dev)
    assert sorted(cols) == ["f", "k", "x"]
    for name, chunks in cols.items():
        got = pa.chunked_array([c.to_arrow() for c in chunks])
        assert got.equals(t.column(name)), name


def test_ipc_file_read_as_rank_shards_and_combined(dev, tmp_path):
    """sharding.read_ipc_shard: two 'ranks' (same process, same GPU — RCCL cannot put two ranks on one device) each read
    their contiguous run of record batches; per-rank statistics combined by the rank-ordered combine (agpu_reduce_combine,
    what agpu_comm_reduce runs after its all-gather) must equal the oracle's sharded result"""
    import ctypes as C

    import oracle as O
    from arrow_gpu_amd import _capi as capi
    from arrow_gpu_amd import sharding
    from arrow_gpu_amd.gpu_utils import ArrowComputePipeline

    n, per_batch = 1_000_000, 65_536
    x = O.synth_f32(n, 5, 0, -1.0, 1.0)
    path = tmp_path / "col.arrow"
    with pa.OSFile(str(path), "wb") as f, pa.ipc.new_file(f, pa.schema([("x", pa.float32())])) as w:
        for s in range(0, n, per_batch):
            w.write_batch(pa.record_batch([pa.array(x[s:s + per_batch])], names=["x"]))
    p = ArrowComputePipeline(dev, "shards")
    world = 2
    parts = [sharding.read_ipc_shard(str(path), dev, r, world)["x"] for r in range(world)]
    assert sum(len(pt) for pt in parts) == (n + per_batch - 1) // per_batch
    # one statistic per (rank, batch) record, in row order: the combine takes any number of records
    stats = {op: [] for op in (capi.RED_SUM, capi.RED_MIN, capi.RED_MAX)}
    chunks = [a for pt in parts for a in pt]
    out = dev.create_empty_buffer(16)
    for op in stats:
        recs = np.zeros(len(chunks) * 2, np.uint64)
        for k, a in enumerate(chunks):
            capi.call("agpu_reduce", p._handle, op, capi.F32, C.c_void_p(a.data.ptr), None, a.len, C.c_void_p(out.ptr))
            recs[2 * k] = int(dev.retrive_data(out, 4, pipeline=p).view(np.uint32)[0])
            recs[2 * k + 1] = a.len
        d = dev.create_gpu_buffer_with_data(recs)
        capi.call("agpu_reduce_combine", p._handle, op, capi.F32, 0, C.c_void_p(d.ptr), len(chunks), C.c_void_p(out.ptr))
        got = dev.retrive_data(out, 4, pipeline=p).view(np.float32)[0]
        exp = O.sharded_reduce(op, O.F32, [x[s:s + per_batch] for s in range(0, n, per_batch)])
        assert np.float32(got).view(np.uint32) == np.float32(exp).view(np.uint32), op


def test_device_columns_to_lz4_compressed_file(dev, tmp_path):
    """write_ipc(compression="lz4") from HBM: download, LZ4 frames, pyarrow reads the table back"""
    import arrow_gpu_amd as ag
    from arrow_gpu_amd.ipc import write_ipc

    n = 300_000
    rng = np.random.default_rng(12)
    t = pa.table({"k": pa.array(np.repeat(np.arange(n // 100, dtype=np.int32), 100), mask=np.arange(n) % 9 == 0),
                  "x": pa.array(rng.standard_normal(n).astype(np.float32)), "b": pa.array(rng.random(n) < 0.1)})
    cols = {name: ag.from_arrow(t.column(name).chunk(0), dev) for name in t.schema.names}
    path = tmp_path / "out.feather"
    assert write_ipc(cols, str(path), file_format=True, compression="lz4") is None
    assert path.stat().st_size < 0.6 * t.nbytes
    back = pa.ipc.open_file(pa.memory_map(str(path))).read_all()
    back.validate(full=True)
    assert back.equals(t)


@pytest.mark.parametrize("compression", [None, "lz4"])
def test_dictionary_encoded_columns_decode_on_the_gpu(dev, compression):
    """Dictionary-encoded numeric columns: read_column uploads indices + dictionary and gathers with agpu_take (nulls'
    unspecified indices neutralised); read_batch goes through the host decode — both must equal pyarrow's decoded column"""
    from arrow_gpu_amd.ipc import IpcReader

    rng = np.random.default_rng(6)
    n = 200_000
    f = pa.array(rng.choice(np.array([0.5, -1.25, 3.0, 1e10, -0.0], np.float32), n), mask=rng.random(n) < 0.15).dictionary_encode()
    i = pa.DictionaryArray.from_arrays(pa.array(rng.integers(0, 300, n).astype(np.int16)), pa.array(np.arange(1000, 1300, dtype=np.int32)))
    garbage = rng.integers(-128, 127, n).astype(np.int8)   # null slots hold arbitrary (even negative) indices
    mask = rng.random(n) < 0.5
    idx8 = np.where(mask, garbage, rng.integers(0, 7, n)).astype(np.int8)
    u = pa.DictionaryArray.from_arrays(pa.Array.from_buffers(pa.int8(), n, [pa.py_buffer(np.packbits(~mask, bitorder="little").tobytes()), pa.py_buffer(idx8)]),
                                       pa.array(np.arange(7, dtype=np.uint16) * 1000))
    allnull = pa.DictionaryArray.from_arrays(pa.array([None] * n, pa.int32()), pa.array([], pa.float32()))
    t = pa.table({"f": f, "i": i, "u": u, "z": allnull, "plain": pa.array(rng.integers(0, 9, n).astype(np.uint8))})
    sink = pa.BufferOutputStream()
    opts = pa.ipc.IpcWriteOptions(compression=compression) if compression else pa.ipc.IpcWriteOptions()
    with pa.ipc.new_file(sink, t.schema, options=opts) as w:
        for b in t.to_batches(max_chunksize=70_000):
            w.write_batch(b)
    batches = t.to_batches(max_chunksize=70_000)
    with IpcReader(sink.getvalue().to_pybytes()) as r:
        for bi, batch in enumerate(batches):
            whole = r.read_batch(bi, dev)  # host decode + table import
            for name in t.schema.names:
                col = batch.column(t.schema.names.index(name))
                dense = col.dictionary_decode() if pa.types.is_dictionary(col.type) else col
                g = r.read_column(bi, r.column_index(name), dev)  # GPU decode for the dictionary columns
                for got in (g.to_arrow(), whole[name].to_arrow()):
                    assert got.null_count == dense.null_count
                    assert got.equals(dense), (bi, name)
