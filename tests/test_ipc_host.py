"""CPU: the C ABI's Arrow IPC reader / writer (csrc/arrow_ipc.hip) against pyarrow's — host-only calls,
no GPU.  pyarrow writes → our reader must see the same schema, rows, values and validity (streaming and file format,
several record batches, unsupported column types in between); our writer → pyarrow must read back equal tables (sliced
inputs, nulls, all nine array types); truncated and corrupted input must be refused, never crash."""
import ctypes as C

import numpy as np
import pytest

pa = pytest.importorskip("pyarrow")

from arrow_gpu_amd import _capi as capi  # noqa: E402
from arrow_gpu_amd.ipc import IpcReader, IpcWriter  # noqa: E402

TYPES = [("f", pa.float32(), capi.F32), ("i", pa.int32(), capi.I32), ("I", pa.uint32(), capi.U32), ("s", pa.int16(), capi.I16),
         ("S", pa.uint16(), capi.U16), ("c", pa.int8(), capi.I8), ("C", pa.uint8(), capi.U8), ("b", pa.bool_(), capi.BOOL),
         ("tdD", pa.date32(), capi.DATE32)]


def make_array(rng, typ, n, null_frac):
    if typ == pa.bool_():
        vals = rng.random(n) < 0.5
    elif typ == pa.float32():
        vals = rng.standard_normal(n).astype(np.float32)
    elif typ == pa.date32():
        vals = rng.integers(-10000, 30000, n).astype(np.int32)
    else:
        info = np.iinfo(typ.to_pandas_dtype())
        vals = rng.integers(info.min, int(info.max) + 1, n).astype(typ.to_pandas_dtype())
    mask = rng.random(n) < null_frac if null_frac else None
    return pa.array(vals, type=typ, mask=mask)


def make_table(rng, n, null_frac=0.2, with_unsupported=True):
    cols, names = [], []
    for k, (fmt, typ, _) in enumerate(TYPES):
        cols.append(make_array(rng, typ, n, null_frac if k % 3 != 2 else 0.0))
        names.append(f"col_{fmt}")
        if with_unsupported and k == 1:
            cols.append(pa.array([None if i % 7 == 0 else "s" * (i % 5) for i in range(n)], type=pa.utf8()))
            names.append("text")
        if with_unsupported and k == 3:
            cols.append(pa.array([[i, i + 1] if i % 3 else None for i in range(n)], type=pa.list_(pa.int32())))
            names.append("lists")
            cols.append(pa.array(rng.integers(0, 1 << 40, n), type=pa.int64()))
            names.append("wide")
        if with_unsupported and k == 5:
            cols.append(pa.array([{"x": i, "y": float(i)} for i in range(n)], type=pa.struct([("x", pa.int32()), ("y", pa.float64())])))
            names.append("rec")
            cols.append(pa.array(["a", "b", "a", None] * (n // 4) + ["a"] * (n % 4)).dictionary_encode())
            names.append("dict")
    return pa.table(cols, names=names)


def serialise(table, file_format, batch_rows):
    sink = pa.BufferOutputStream()
    w = (pa.ipc.new_file if file_format else pa.ipc.new_stream)(sink, table.schema)
    for b in table.to_batches(max_chunksize=batch_rows):
        w.write_batch(b)
    w.close()
    return sink.getvalue().to_pybytes()


def unpack_bits(bits, n):
    return np.unpackbits(np.asarray(bits), bitorder="little")[:n].astype(bool)


@pytest.mark.parametrize("file_format", [False, True])
@pytest.mark.parametrize("n,batch_rows", [(1000, 1000), (1003, 257), (0, 10), (5, 1)])
def test_reader_sees_what_pyarrow_wrote(file_format, n, batch_rows):
    rng = np.random.default_rng(n + batch_rows)
    table = make_table(rng, n)
    data = serialise(table, file_format, batch_rows)
    batches = table.to_batches(max_chunksize=batch_rows)
    with IpcReader(data) as r:
        assert [f.name for f in r.fields] == table.schema.names
        by_name = {f.name: f for f in r.fields}
        for fmt, _, code in TYPES:
            assert by_name[f"col_{fmt}"].dtype == code and by_name[f"col_{fmt}"].format == fmt
        for name in ("text", "lists", "wide", "rec", "dict"):
            assert by_name[name].dtype == -1
        assert by_name["text"].format == "u" and by_name["wide"].format == "l"
        assert r.num_batches == len(batches)
        for bi, batch in enumerate(batches):
            assert r.batch_rows(bi) == batch.num_rows
            for ci, f in enumerate(r.fields):
                if f.dtype < 0:
                    with pytest.raises(capi.OperationNotSupported):
                        r.column_view(bi, ci)
                    continue
                values, validity, length, nulls = r.column_view(bi, ci)
                col = batch.column(ci)
                assert length == len(col) and nulls == col.null_count
                valid = np.ones(length, bool) if validity is None else unpack_bits(validity, length)
                assert (validity is None) == (col.null_count == 0)
                exp_valid = np.asarray(col.is_valid())
                assert np.array_equal(valid, exp_valid)
                if f.dtype == capi.BOOL:
                    got = unpack_bits(values, length)
                    exp = np.asarray(col.fill_null(False))
                else:
                    got = np.asarray(values)
                    storage = col.cast(pa.int32()) if col.type == pa.date32() else col
                    exp = storage.fill_null(0).to_numpy(zero_copy_only=False).astype(got.dtype)
                assert np.array_equal(got[exp_valid], exp[exp_valid])


def host_write(table_batches, schema, file_format, sink=None):
    codes = {t: c for _, t, c in TYPES}
    w = IpcWriter([(f.name, codes[f.type], f.nullable) for f in schema], sink, file_format)
    for b in table_batches:
        w.write_host_batch(b.columns)
    return w.finish()


@pytest.mark.parametrize("file_format", [False, True])
@pytest.mark.parametrize("n,batch_rows,offset", [(1000, 1000, 0), (1003, 257, 0), (0, 10, 0), (777, 100, 13), (64, 64, 8), (9, 3, 1)])
def test_pyarrow_reads_what_the_writer_wrote(file_format, n, batch_rows, offset):
    rng = np.random.default_rng(7 * n + batch_rows + offset)
    table = make_table(rng, n + offset, with_unsupported=False).slice(offset)  # offset != 0: sliced arrays, bit offsets in the bitmaps
    batches = table.to_batches(max_chunksize=batch_rows)
    data = host_write(batches, table.schema, file_format)
    rd = (pa.ipc.open_file if file_format else pa.ipc.open_stream)(pa.BufferReader(data))
    got = rd.read_all()
    assert got.schema.names == table.schema.names
    assert [f.type for f in got.schema] == [f.type for f in table.schema]
    assert got.num_rows == table.num_rows
    if file_format:
        assert rd.num_record_batches == len(batches)
    for name in table.schema.names:
        a, b = got.column(name).combine_chunks(), table.column(name).combine_chunks()
        assert a.null_count == b.null_count
        assert a.equals(b), name
    got.validate(full=True)
    # and our own reader agrees with pyarrow's about the bytes we wrote
    with IpcReader(data) as r:
        assert r.num_batches == len(batches) and [f.name for f in r.fields] == table.schema.names
        assert sum(r.batch_rows(i) for i in range(r.num_batches)) == table.num_rows


def test_writer_to_a_file_descriptor_and_mmap_reader(tmp_path):
    rng = np.random.default_rng(5)
    table = make_table(rng, 5000, with_unsupported=False)
    path = tmp_path / "cols.arrow"
    assert host_write(table.to_batches(max_chunksize=1200), table.schema, True, sink=str(path)) is None
    with pa.memory_map(str(path)) as src:
        assert pa.ipc.open_file(src).read_all().equals(table)
    with IpcReader(str(path)) as r:  # memory-mapped
        assert r.num_batches == 5
        values, validity, length, nulls = r.column_view(4, r.column_index("col_f"))
        exp = table.column("col_f").chunk(0)[4800:]
        assert length == 200 and nulls == exp.null_count
        ok = np.asarray(exp.is_valid())
        assert np.array_equal(np.asarray(values)[ok], exp.fill_null(0).to_numpy(zero_copy_only=False)[ok])


def test_no_nulls_means_no_validity_buffer():
    t = pa.table({"a": pa.array([1, 2, 3], pa.int32()), "b": pa.array([True, None, False])})
    data = host_write(t.to_batches(), t.schema, False)
    with IpcReader(data) as r:
        _, validity, _, nulls = r.column_view(0, 0)
        assert validity is None and nulls == 0
        _, validity, _, nulls = r.column_view(0, 1)
        assert nulls == 1 and unpack_bits(validity, 3).tolist() == [True, False, True]


@pytest.mark.parametrize("file_format", [False, True])
def test_lz4_compressed_batches_are_decompressed(file_format):
    """BodyCompression LZ4_FRAME (pyarrow's `compression="lz4"`, Feather V2's default): every buffer carries its
    uncompressed length and an LZ4 frame; the reader decodes the frames itself."""
    rng = np.random.default_rng(21)
    n = 50_000
    cols = {
        "runs": pa.array(np.repeat(np.arange(n // 100, dtype=np.int32), 100)),                 # long matches, linked blocks
        "noise": pa.array(rng.standard_normal(n).astype(np.float32), mask=rng.random(n) < 0.1),   # incompressible: stored blocks / −1 prefix
        "small": pa.array(rng.integers(0, 3, n).astype(np.uint8)),
        "flags": pa.array(rng.random(n) < 0.02),
        "text": pa.array(["x" * (i % 7) for i in range(n)]),                                      # skipped, still compressed
        "dates": pa.array(np.arange(n, dtype=np.int32) // 30, type=pa.date32(), mask=np.arange(n) % 11 == 0),
    }
    table = pa.table(cols)
    try:
        opts = pa.ipc.IpcWriteOptions(compression="lz4")
    except Exception:
        pytest.skip("pyarrow built without lz4")
    sink = pa.BufferOutputStream()
    with (pa.ipc.new_file if file_format else pa.ipc.new_stream)(sink, table.schema, options=opts) as w:
        for b in table.to_batches(max_chunksize=20_000):
            w.write_batch(b)
    data = sink.getvalue().to_pybytes()
    assert len(data) < table.nbytes  # it really is compressed
    batches = table.to_batches(max_chunksize=20_000)
    with IpcReader(data) as r:
        assert r.num_batches == len(batches)
        for bi, batch in enumerate(batches):
            for ci, f in enumerate(r.fields):
                if f.dtype < 0:
                    continue
                values, validity, length, nulls = r.column_view(bi, ci)
                col = batch.column(ci)
                assert length == len(col) and nulls == col.null_count
                ok = np.asarray(col.is_valid())
                if validity is not None:
                    assert np.array_equal(unpack_bits(validity, length), ok)
                if f.dtype == capi.BOOL:
                    got, exp = unpack_bits(values, length), np.asarray(col.fill_null(False))
                else:
                    storage = col.cast(pa.int32()) if col.type == pa.date32() else col
                    got, exp = np.asarray(values), storage.fill_null(0).to_numpy(zero_copy_only=False)
                assert np.array_equal(got[ok], exp[ok]), (bi, f.name)


def test_feather_v2_file_with_default_compression(tmp_path):
    import pyarrow.feather as feather

    rng = np.random.default_rng(4)
    t = pa.table({"a": pa.array(rng.integers(0, 50, 100_000).astype(np.int32)), "b": pa.array(rng.random(100_000).astype(np.float32))})
    path = tmp_path / "t.feather"
    import warnings

    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", FutureWarning)  # pyarrow ≥ 24 deprecates the feather module; the FORMAT is the IPC file format
            feather.write_feather(t, str(path))  # compression="lz4" by default when available
    except Exception:
        pytest.skip("feather / lz4 not available")
    with IpcReader(str(path)) as r:
        got = {f.name: np.concatenate([np.asarray(r.column_view(b, i)[0]) for b in range(r.num_batches)]) for i, f in enumerate(r.fields)}
    assert np.array_equal(got["a"], t.column("a").to_numpy()) and np.array_equal(got["b"], t.column("b").to_numpy())


def test_compressed_and_mismatched_input_is_refused():
    t = pa.table({"a": pa.array(np.arange(1000, dtype=np.int32))})
    sink = pa.BufferOutputStream()
    try:
        opts = pa.ipc.IpcWriteOptions(compression="zstd")
    except Exception:
        pytest.skip("pyarrow built without zstd")
    with pa.ipc.new_stream(sink, t.schema, options=opts) as w:
        w.write_table(t)
    with IpcReader(sink.getvalue().to_pybytes()) as r:
        assert r.fields[0].dtype == capi.I32 and r.num_batches == 1
        with pytest.raises(capi.OperationNotSupported):  # ZSTD: not decoded here
            r.column_view(0, 0)
    w = IpcWriter([("a", capi.I32, True), ("b", capi.F32, True)])
    with pytest.raises(capi.ArrowErrorGPU):  # columns of different length
        w.write_host_batch([pa.array([1, 2, 3], pa.int32()), pa.array([1.0], pa.float32())])
    with pytest.raises(capi.OperationNotSupported):
        IpcWriter([("a", 99, True)])
    # a FieldNode length far beyond the body (len × 4 would wrap around 2^64) must be refused, not believed
    small = pa.table({"a": pa.array(np.arange(7, dtype=np.int32))})
    sink = pa.BufferOutputStream()
    with pa.ipc.new_stream(sink, small.schema) as w2:
        w2.write_table(small)
    raw = bytearray(sink.getvalue().to_pybytes())
    seven = (7).to_bytes(8, "little")
    hits = [i for i in range(len(raw) - 8) if raw[i:i + 8] == seven]
    assert hits
    for i in hits:  # RecordBatch.length and FieldNode.length both hold 7
        raw[i:i + 8] = (1 << 62).to_bytes(8, "little")
    with IpcReader(bytes(raw)) as r:
        with pytest.raises(capi.ArrowErrorGPU):
            r.column_view(0, 0)
    with pytest.raises(capi.ArrowErrorGPU):
        IpcReader(b"definitely not arrow")
    with pytest.raises(capi.ArrowErrorGPU):
        IpcReader(b"ARROW1\0\0" + b"\0" * 40)


@pytest.mark.parametrize("file_format", [False, True])
def test_truncated_and_corrupted_input_never_crashes(file_format):
    rng = np.random.default_rng(11)
    table = make_table(rng, 300)
    data = serialise(table, file_format, 100)
    opened = refused = 0
    for cut in list(range(0, 600, 7)) + list(range(600, len(data), 211)):
        try:
            with IpcReader(data[:cut] if cut else b"\0") as r:
                for b in range(r.num_batches):
                    for c, f in enumerate(r.fields):
                        if f.dtype >= 0:
                            r.column_view(b, c)
            opened += 1
        except capi.ArrowErrorGPU:
            refused += 1
    assert refused > 0
    if not file_format:  # the end-of-stream marker is optional: a stream that simply stops after a message is complete
        with IpcReader(data[:-8]) as r:
            assert r.num_batches == 3
    # byte flips inside the metadata of the schema and the first record batch
    base = bytearray(data)
    for trial in range(400):
        buf = bytearray(base)
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(0, min(len(buf), 2500))) if not file_format or trial % 2 else int(rng.integers(max(0, len(buf) - 1500), len(buf)))
            buf[pos] = int(rng.integers(0, 256))
        try:
            with IpcReader(bytes(buf)) as r:
                for b in range(r.num_batches):
                    r.batch_rows(b)
                    for c, f in enumerate(r.fields):
                        if f.dtype >= 0:
                            try:
                                values, validity, n, _ = r.column_view(b, c)
                                if len(values):
                                    int(np.asarray(values).view(np.uint8)[-1])  # touch the last byte: must be inside the source
                            except capi.ArrowErrorGPU:
                                pass
        except capi.ArrowErrorGPU:
            pass


@pytest.mark.parametrize("file_format", [False, True])
def test_lz4_compressing_writer_is_read_by_pyarrow_and_by_us(file_format):
    """agpu_ipc_writer_set_compression(1): every buffer {uncompressed length, LZ4 frame} (or {-1, bytes} when a frame would
    not be smaller); pyarrow's reader (liblz4, which verifies the frame header checksum) must return the source table"""
    rng = np.random.default_rng(31)
    n = 120_000
    t = pa.table({
        "runs": pa.array(np.repeat(np.arange(n // 200, dtype=np.int32), 200), mask=np.arange(n) % 17 == 0),
        "noise": pa.array(rng.standard_normal(n).astype(np.float32)),
        "small": pa.array(rng.integers(0, 4, n).astype(np.uint8), mask=rng.random(n) < 0.3),
        "flags": pa.array(rng.random(n) < 0.01),
        "days": pa.array((np.arange(n) // 1000).astype(np.int32), type=pa.date32()),
    })
    codes = {tp: c for _, tp, c in TYPES}
    w = IpcWriter([(f.name, codes[f.type], True) for f in t.schema], None, file_format, compression="lz4")
    batches = t.slice(3).to_batches(max_chunksize=50_000)  # sliced inputs too
    for b in batches:
        w.write_host_batch(b.columns)
    data = w.finish()
    assert len(data) < 0.7 * t.nbytes
    got = (pa.ipc.open_file if file_format else pa.ipc.open_stream)(pa.BufferReader(data)).read_all()
    got.validate(full=True)
    assert got.equals(t.slice(3))
    with IpcReader(data) as r:
        assert r.num_batches == len(batches)
        for bi, b in enumerate(batches):
            v, _, ln, nulls = r.column_view(bi, 0)
            col = b.column(0)
            ok = np.asarray(col.is_valid())
            assert ln == len(col) and nulls == col.null_count
            assert np.array_equal(np.asarray(v)[ok], col.fill_null(0).to_numpy(zero_copy_only=False)[ok])
    with pytest.raises(capi.OperationNotSupported):
        IpcWriter([("a", capi.I32, True)], compression="zstd")


@pytest.mark.parametrize("file_format", [False, True])
@pytest.mark.parametrize("compression", [None, "lz4"])
def test_dictionary_encoded_numeric_columns_are_decoded(file_format, compression):
    """Dictionary-encoded columns whose VALUES have a GPU array type (f32 / i32 / u16 … with int8 / int16 / int32 indices)
    read as their decoded values; string dictionaries stay unsupported; nulls travel in the index validity"""
    rng = np.random.default_rng(5)
    n = 40_000
    f = pa.array(rng.choice(np.array([0.5, -1.25, 3.0, 1e10, -0.0], np.float32), n), mask=rng.random(n) < 0.15).dictionary_encode()
    i = pa.DictionaryArray.from_arrays(pa.array(rng.integers(0, 300, n).astype(np.int16)), pa.array(np.arange(1000, 1300, dtype=np.int32)))
    u = pa.DictionaryArray.from_arrays(pa.array(rng.integers(0, 7, n).astype(np.int8), mask=rng.random(n) < 0.5),
                                       pa.array(np.arange(7, dtype=np.uint16) * 1000))
    s_ = pa.array(rng.choice(["a", "bb", "ccc"], n)).dictionary_encode()
    plain = pa.array(rng.integers(0, 100, n).astype(np.int32))
    t = pa.table({"f": f, "i": i, "s": s_, "u": u, "plain": plain})
    sink = pa.BufferOutputStream()
    opts = pa.ipc.IpcWriteOptions(compression=compression) if compression else pa.ipc.IpcWriteOptions()
    with (pa.ipc.new_file if file_format else pa.ipc.new_stream)(sink, t.schema, options=opts) as w:
        for b in t.to_batches(max_chunksize=15_000):
            w.write_batch(b)
    data = sink.getvalue().to_pybytes()
    batches = t.to_batches(max_chunksize=15_000)
    with IpcReader(data) as r:
        by = {fl.name: fl for fl in r.fields}
        assert (by["f"].dtype, by["i"].dtype, by["u"].dtype, by["plain"].dtype, by["s"].dtype) == (capi.F32, capi.I32, capi.U16, capi.I32, -1)
        assert r.num_batches == len(batches)
        for bi, batch in enumerate(batches):
            for name in ("f", "i", "u", "plain"):
                ci = r.column_index(name)
                values, validity, length, nulls = r.column_view(bi, ci)
                col = batch.column(ci)
                dense = col.dictionary_decode() if pa.types.is_dictionary(col.type) else col
                ok = np.asarray(dense.is_valid())
                assert length == len(col) and nulls == col.null_count
                if validity is not None:
                    assert np.array_equal(unpack_bits(validity, length), ok)
                exp = dense.fill_null(0).to_numpy(zero_copy_only=False)
                assert np.array_equal(np.asarray(values)[ok].view(np.uint8), exp[ok].view(np.uint8)), (bi, name)
            with pytest.raises(capi.OperationNotSupported):
                r.column_view(bi, r.column_index("s"))


def test_dictionary_deltas_and_replacements_in_a_stream():
    """A stream may replace a dictionary between record batches or extend it with a delta batch; every record batch is
    decoded with the dictionary in force where it stands"""
    vals = [np.array([10, 20, 30], np.int32), np.array([10, 20, 30, 40, 50], np.int32), np.array([7, 8], np.int32)]  # extend, then replace
    idx = [np.array([0, 2, 1, 1], np.int8), np.array([4, 0, 3, 3, 2], np.int8), np.array([1, 1, 0], np.int8)]
    typ = pa.dictionary(pa.int8(), pa.int32())
    batches = [pa.record_batch([pa.DictionaryArray.from_arrays(pa.array(i), pa.array(v))], schema=pa.schema([("d", typ)])) for i, v in zip(idx, vals)]
    sink = pa.BufferOutputStream()
    with pa.ipc.new_stream(sink, batches[0].schema, options=pa.ipc.IpcWriteOptions(emit_dictionary_deltas=True)) as w:
        for b in batches:
            w.write_batch(b)
    with IpcReader(sink.getvalue().to_pybytes()) as r:
        assert r.num_batches == 3 and r.fields[0].dtype == capi.I32
        for k in range(3):
            values, validity, length, nulls = r.column_view(k, 0)
            assert validity is None and nulls == 0
            assert np.array_equal(np.asarray(values), vals[k][idx[k]])


# ---- a tiny schema whose struct Fields list K children that all ARE one shared next-level table
def shared_child_schema_stream(depth: int, k: int) -> bytes:
    """An Arrow IPC stream holding one Schema message, written by hand: Field level d is a Struct whose `children` vector
    has `k` entries that all point at the single Field table of level d + 1; the last level is an Int32.  Valid
    Flatbuffers (offsets only point forward) — a naive depth-first layout walk costs k^depth visits."""
    import struct

    buf = bytearray()

    def align(a):
        while len(buf) % a:
            buf.append(0)

    def table(fields):
        """fields: list of (kind, value) per vtable slot — kind in {None, 'u8', 'i16', 'i64', 'off'}; returns (table_pos, {slot: pos})
        with offset slots left 0 for patching."""
        sizes = {"u8": 1, "i16": 2, "i64": 8, "off": 4}
        layout, off = {}, 4
        for i, (kind, _) in enumerate(fields):
            if kind is None:
                continue
            sz = sizes[kind]
            off = (off + sz - 1) // sz * sz
            layout[i] = off
            off += sz
        tab_size = (off + 3) // 4 * 4
        vt = struct.pack("<HH", 4 + 2 * len(fields), tab_size) + b"".join(struct.pack("<H", layout.get(i, 0)) for i in range(len(fields)))
        align(8)
        if (len(buf) + len(vt)) % 8:  # keep the table itself 8-aligned (i64 members)
            buf.extend(b"\0" * (8 - (len(buf) + len(vt)) % 8))
        vpos = len(buf)
        buf.extend(vt)
        tpos = len(buf)
        body = bytearray(tab_size)
        struct.pack_into("<i", body, 0, tpos - vpos)
        for i, (kind, val) in enumerate(fields):
            if kind in ("u8",):
                struct.pack_into("<B", body, layout[i], val)
            elif kind == "i16":
                struct.pack_into("<h", body, layout[i], val)
            elif kind == "i64":
                struct.pack_into("<q", body, layout[i], val)
        buf.extend(body)
        return tpos, {i: tpos + o for i, o in layout.items()}

    def patch(slot_pos, target):
        struct.pack_into("<I", buf, slot_pos, target - slot_pos)

    buf.extend(b"\0\0\0\0")  # root offset
    # Message {version: V5 = 4, header_type: Schema = 1, header, bodyLength}
    msg, m = table([("i16", 4), ("u8", 1), ("off", 0), ("i64", 0)])
    struct.pack_into("<I", buf, 0, msg)
    schema, s = table([("i16", 0), ("off", 0)])
    patch(m[2], schema)
    align(4)
    fvec = len(buf)
    buf.extend(struct.pack("<II", 1, 0))
    patch(s[1], fvec)
    prev_slots = [fvec + 4]
    for d in range(depth + 1):
        leaf = d == depth
        # Field {name, nullable, type_type, type, dictionary, children}
        f, fs = table([(None, 0), ("u8", 1), ("u8", 2 if leaf else 13), ("off", 0), (None, 0), (None, 0) if leaf else ("off", 0)])
        for sp in prev_slots:
            patch(sp, f)
        if leaf:
            ty, _ = table([("i16", 0)])  # Int {bitWidth: i32} — written below as a 4-byte member
            # rewrite as a proper Int table: bitWidth (i32) = 32, is_signed (bool) = 1
            del buf[ty - 6:]
            align(4)
            vt = struct.pack("<HHHH", 8, 12, 4, 8)
            if (len(buf) + len(vt)) % 4:
                buf.extend(b"\0" * (4 - (len(buf) + len(vt)) % 4))
            vpos = len(buf)
            buf.extend(vt)
            ty = len(buf)
            buf.extend(struct.pack("<iiB3x", ty - vpos, 32, 1))
            patch(fs[3], ty)
        else:
            ty, _ = table([])  # Struct_ {}
            patch(fs[3], ty)
            align(4)
            cvec = len(buf)
            buf.extend(struct.pack("<I", k) + b"\0" * (4 * k))
            patch(fs[5], cvec)
            prev_slots = [cvec + 4 + 4 * i for i in range(k)]
    while len(buf) % 8:
        buf.append(0)
    return struct.pack("<II", 0xFFFFFFFF, len(buf)) + bytes(buf) + struct.pack("<II", 0xFFFFFFFF, 0)


def test_shared_child_tables_do_not_blow_up_the_schema_walk():
    """k^depth visits when walked naively (592 bytes took 14 ms, 736 bytes 919 ms, ~1 KB hours): now linear — every Field
    table is laid out once — and a schema that describes more than 2^31 nodes is refused as malformed"""
    import time

    lib = capi.lib()
    lib.agpu_ipc_close.restype = None
    for depth, k, expect_ok in ((3, 4, True), (10, 4, True), (13, 4, True), (40, 4, False), (60, 16, False), (80, 4, True)):
        blob = shared_child_schema_stream(depth, k)
        r = C.c_void_p()
        t0 = time.perf_counter()
        st = lib.agpu_ipc_open(blob, len(blob), C.byref(r))
        dt = time.perf_counter() - t0
        assert dt < 0.5, (depth, k, dt, len(blob))
        if expect_ok:
            assert st == capi.OK, (depth, k, st, lib.agpu_last_error())
            n = C.c_int32()
            assert lib.agpu_ipc_num_fields(r, C.byref(n)) == capi.OK and n.value == 1
            lib.agpu_ipc_close(r)
        else:
            assert st == capi.ERR_SHAPE, (depth, k, st)
    # a sane nested schema written by pyarrow still opens and skips the struct column correctly
    pa = pytest.importorskip("pyarrow")
    t = pa.table({"s": pa.array([{"a": 1, "b": 2.0}] * 5), "x": pa.array(np.arange(5, dtype=np.int32))})
    with IpcReader(serialise(t, False, 5)) as rd:
        got = rd.column_view(0, 1)
        assert list(got[0]) == [0, 1, 2, 3, 4]
