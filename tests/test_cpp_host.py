"""The C++ host layer (host/arrow_gpu.hpp — the compiled-language mirror of the reference's Rust API) builds against
the C ABI and, on the GPU box, passes a port of examples/simple.rs plus a cross-section of the reference's vectors."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_host_api.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "build", "test_host_api")
LIBDIR = os.path.join(ROOT, "arrow_gpu_amd", "lib")


def build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    deps = [SRC, os.path.join(ROOT, "host", "arrow_gpu.hpp"), os.path.join(ROOT, "include", "arrow_gpu.h")]
    if os.path.exists(EXE) and all(os.path.getmtime(EXE) >= os.path.getmtime(d) for d in deps):
        return
    cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-Wall", "-x", "c++", SRC, "-o", EXE, f"-L{LIBDIR}", "-larrow_gpu_hip",
           "-Wl,-rpath," + LIBDIR]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]


def test_cpp_host_layer_compiles_and_fails_loudly_without_gpu():
    import torch

    build()
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present; the run is covered by the gpu-marked test")
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "no device" in r.stdout  # no CPU fallback in the C++ layer either


@pytest.mark.gpu
def test_cpp_host_layer_on_gpu():
    build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


# ---- the C ABI from plain C (what a cgo / JNI / Rust-FFI shim binds): the header must be valid strict C11
C_SRC = os.path.join(ROOT, "examples", "simple.c")
C_EXE = os.path.join(ROOT, "tests", "cpp", "build", "simple_c")


def build_c():
    os.makedirs(os.path.dirname(C_EXE), exist_ok=True)
    cmd = ["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", C_SRC, "-I" + os.path.join(ROOT, "include"),
           f"-L{LIBDIR}", "-larrow_gpu_hip", "-Wl,-rpath," + LIBDIR, "-o", C_EXE]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]


def test_c_example_compiles_as_strict_c11_and_fails_loudly_without_gpu():
    import torch

    build_c()
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present; the run is covered by the gpu-marked test")
    r = subprocess.run([C_EXE], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_c_example_on_gpu():
    build_c()
    r = subprocess.run([C_EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "simple.c OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


# ---- config 5 through the C++ host alone: one thread per GPU, RCCL final reduce (examples/sharded_stats.cpp)
S_SRC = os.path.join(ROOT, "examples", "sharded_stats.cpp")
S_EXE = os.path.join(ROOT, "tests", "cpp", "build", "sharded_stats")


def build_sharded():
    os.makedirs(os.path.dirname(S_EXE), exist_ok=True)
    deps = [S_SRC, os.path.join(ROOT, "host", "arrow_gpu.hpp"), os.path.join(ROOT, "include", "arrow_gpu.h")]
    if os.path.exists(S_EXE) and all(os.path.getmtime(S_EXE) >= os.path.getmtime(d) for d in deps):
        return
    cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O2", "-Wall", "-x", "c++", S_SRC, "-o", S_EXE, f"-L{LIBDIR}", "-larrow_gpu_hip",
           "-Wl,-rpath," + LIBDIR, "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]


def test_sharded_stats_example_compiles():
    build_sharded()


@pytest.mark.gpu
def test_sharded_stats_example_matches_the_sharded_spec():
    import json

    import numpy as np

    import oracle as O

    build_sharded()
    rows = 1 << 24  # 256^3 rows per shard: the sharded f32 Sum equals the reference's whole-column tree bit for bit
    r = subprocess.run([S_EXE, str(rows)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"exit {r.returncode}: " + r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    world = line["world"]
    assert line["identical_on_all_ranks"] is True and world >= 1 and line["scaling"] == "weak"
    assert line["value_GBps"] > 0 and line["add_ms"]["min"] > 0 and line["eq_ms"]["max"] >= line["eq_ms"]["min"] > 0
    assert "librccl" in line["runtime"]
    assert line["one_pass_identical"] is True and line["ms_one_pass_statistics"] > 0  # stats_sharded_op: the three statistics from ONE read of the shard
    shards = [O.synth_f32(rows, 20250418, k * rows, -1000.0, 1000.0) for k in range(world)]
    exp = O.sharded_reduce(O.RED_SUM, O.F32, shards)
    assert np.float32(line["sum"]).view(np.uint32) == np.float32(exp).view(np.uint32), (line["sum"], exp)
    assert np.float32(line["min"]) == min(s.min() for s in shards) and np.float32(line["max"]) == max(s.max() for s in shards)
    if world == 1:
        assert np.float32(exp).view(np.uint32) == np.float32(O.reduce(O.RED_SUM, O.F32, shards[0])).view(np.uint32)
    # strong mode: the same column cut into `world` shards (ragged total: the last shard is short)
    total = 3_000_001
    r = subprocess.run([S_EXE, str(total), str(world), "strong", "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["scaling"] == "strong" and line["rows_total"] == total and line["steps"] == 3
    whole = O.synth_f32(total, 20250418, 0, -1000.0, 1000.0)
    cuts = [(s.row0, s.row0 + s.rows) for s in __import__("arrow_gpu_amd.sharding", fromlist=["x"]).all_shards(total, world)]
    exp = O.sharded_reduce(O.RED_SUM, O.F32, [whole[a:b] for a, b in cuts])
    assert np.float32(line["sum"]).view(np.uint32) == np.float32(exp).view(np.uint32)
    assert np.float32(line["min"]) == whole.min() and np.float32(line["max"]) == whole.max()


# ---- Arrow C Data Interface from plain C (examples/arrow_cdata.c)
A_SRC = os.path.join(ROOT, "examples", "arrow_cdata.c")
A_EXE = os.path.join(ROOT, "tests", "cpp", "build", "arrow_cdata")


def build_arrow_c():
    os.makedirs(os.path.dirname(A_EXE), exist_ok=True)
    cmd = ["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", A_SRC, "-I" + os.path.join(ROOT, "include"),
           f"-L{LIBDIR}", "-larrow_gpu_hip", "-Wl,-rpath," + LIBDIR, "-o", A_EXE]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]


def test_arrow_cdata_example_compiles_as_strict_c11():
    build_arrow_c()


@pytest.mark.gpu
def test_arrow_cdata_example_on_gpu():
    build_arrow_c()
    r = subprocess.run([A_EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "arrow_cdata OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


# ---- overlapped ingest / compute / egress from a compiled host (examples/overlap_pipeline.cpp)
O_SRC = os.path.join(ROOT, "examples", "overlap_pipeline.cpp")
O_EXE = os.path.join(ROOT, "tests", "cpp", "build", "overlap_pipeline")


def build_overlap():
    os.makedirs(os.path.dirname(O_EXE), exist_ok=True)
    if os.path.exists(O_EXE) and os.path.getmtime(O_EXE) >= max(os.path.getmtime(O_SRC), os.path.getmtime(os.path.join(ROOT, "include", "arrow_gpu.h"))):
        return
    cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O2", "-Wall", "-x", "c++", O_SRC, "-o", O_EXE, f"-L{LIBDIR}", "-larrow_gpu_hip",
           "-Wl,-rpath," + LIBDIR, "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]


def test_overlap_pipeline_example_compiles():
    build_overlap()


@pytest.mark.gpu
def test_overlap_pipeline_example_on_gpu():
    import json

    build_overlap()
    r = subprocess.run([O_EXE, str(50_000_003), str(1 << 22)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["ok"] is True and line["rows"] == 50_000_003
    print(r.stdout.strip())


# ---- Arrow IPC file → HBM → kernels → Arrow IPC file from the C++ host (examples/ipc_roundtrip.cpp)
I_SRC = os.path.join(ROOT, "examples", "ipc_roundtrip.cpp")
I_EXE = os.path.join(ROOT, "tests", "cpp", "build", "ipc_roundtrip")


def build_ipc():
    os.makedirs(os.path.dirname(I_EXE), exist_ok=True)
    deps = [I_SRC, os.path.join(ROOT, "host", "arrow_gpu.hpp"), os.path.join(ROOT, "include", "arrow_gpu.h")]
    if os.path.exists(I_EXE) and all(os.path.getmtime(I_EXE) >= os.path.getmtime(d) for d in deps):
        return
    cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O2", "-Wall", "-x", "c++", I_SRC, "-o", I_EXE, f"-L{LIBDIR}", "-larrow_gpu_hip",
           "-Wl,-rpath," + LIBDIR]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]


def _ipc_input(path, n, batch_rows):
    import numpy as np
    import pyarrow as pa

    rng = np.random.default_rng(n)
    a, b = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    k, m = rng.integers(0, 3, n).astype(np.int32), rng.integers(0, 3, n).astype(np.int32)
    ma, mb = rng.random(n) < 0.15, rng.random(n) < 0.15
    t = pa.table({"a": pa.array(a, mask=ma), "note": pa.array([f"r{i % 10}" for i in range(n)]), "b": pa.array(b, mask=mb),
                  "k": pa.array(k, mask=mb), "wide": pa.array(np.arange(n, dtype=np.int64)), "m": pa.array(m)})
    with pa.OSFile(str(path), "wb") as f, pa.ipc.new_file(f, t.schema) as w:
        for rb in t.to_batches(max_chunksize=batch_rows):
            w.write_batch(rb)
    return a, b, k, m, ma, mb


def test_ipc_example_reads_the_schema_without_a_gpu(tmp_path):
    import json

    pytest.importorskip("pyarrow")
    build_ipc()
    _ipc_input(tmp_path / "in.arrow", 1000, 300)
    r = subprocess.run([I_EXE, "--schema", str(tmp_path / "in.arrow")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    info = json.loads(r.stdout)
    assert [(f["name"], f["format"], f["dtype"]) for f in info["fields"]] == [
        ("a", "f", 1), ("note", "u", -1), ("b", "f", 1), ("k", "i", 5), ("wide", "l", -1), ("m", "i", 5)]
    assert info["batch_rows"] == [300, 300, 300, 100]


@pytest.mark.gpu
def test_ipc_example_on_gpu(tmp_path):
    import json

    import numpy as np

    import oracle as O

    pa = pytest.importorskip("pyarrow")
    build_ipc()
    n = 1_000_003
    a, b, k, m, ma, mb = _ipc_input(tmp_path / "in.arrow", n, 250_000)
    r = subprocess.run([I_EXE, str(tmp_path / "in.arrow"), str(tmp_path / "out.arrow")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["ok"] is True and line["rows"] == n and line["batches"] == 5
    got = pa.ipc.open_file(pa.memory_map(str(tmp_path / "out.arrow"))).read_all()
    got.validate(full=True)
    s, e = got.column("sum").combine_chunks(), got.column("eq").combine_chunks()
    vs, ve = ~(ma | mb), ~mb
    assert np.array_equal(np.asarray(s.is_valid()), vs) and np.array_equal(np.asarray(e.is_valid()), ve)
    exp = O.binary(O.OP_ADD, O.F32, a, b)
    assert np.array_equal(s.fill_null(0).to_numpy(zero_copy_only=False).view(np.uint32)[vs], exp.view(np.uint32)[vs])
    assert np.array_equal(np.asarray(e.fill_null(False))[ve], (k == m)[ve])


# ---- the reference's criterion shapes from the C++ host, the value on the host at the end of every iteration (examples/latency.cpp)
L_SRC = os.path.join(ROOT, "examples", "latency.cpp")
L_EXE = os.path.join(ROOT, "tests", "cpp", "build", "latency")


def build_latency():
    os.makedirs(os.path.dirname(L_EXE), exist_ok=True)
    deps = [L_SRC, os.path.join(ROOT, "host", "arrow_gpu.hpp"), os.path.join(ROOT, "include", "arrow_gpu.h")]
    if os.path.exists(L_EXE) and all(os.path.getmtime(L_EXE) >= os.path.getmtime(d) for d in deps):
        return
    cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O2", "-Wall", "-x", "c++", L_SRC, "-o", L_EXE, f"-L{LIBDIR}", "-larrow_gpu_hip",
           "-Wl,-rpath," + LIBDIR]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]


def test_latency_example_compiles():
    build_latency()


@pytest.mark.gpu
def test_latency_example_returns_the_sums():
    """the program checks every sum it reads back (2 n); the times are reported, not asserted — except that a value on the host must not cost a
    millisecond (a wait that fell through to a blocking path on every iteration would)"""
    import json

    build_latency()
    r = subprocess.run([L_EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, f"exit {r.returncode}: " + r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    for k in ("u32_sum_1Mi_value_on_host_us", "u32_sum_10Mi_value_on_host_us", "u32_sum_1Mi_device_sync_us", "f32_add_scalar_10Mi_device_sync_us"):
        assert 1.0 < line[k]["best"] <= line[k]["median"] < 1000.0, (k, line[k])
    print(json.dumps(line))
