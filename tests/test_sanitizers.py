"""CPU: AddressSanitizer + UBSan over the host-side C of the repo (SURVEY §5 — the reference has no sanitizers and
relies on WGSL robust buffer access; GPU ASan is not available on the pool, so the device code is covered by the
guard/tail tests instead).  The oracle (oracle/agpu_oracle.c) and the CPU baseline (oracle/cpu_baseline.c) are rebuilt
with -fsanitize=address,undefined -fno-sanitize-recover=all and driven by the very same Python harnesses — the golden
runner over every reference vector, and the baseline check — in a subprocess with libasan preloaded."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "oracle", "_build")


def _asan_env():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan.so not found next to gcc")
    env = dict(os.environ)
    env.update({"LD_PRELOAD": libasan, "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1:halt_on_error=1",
                "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1",
                "ORACLE_LIB": os.path.join(BUILD, "liboracle_asan.so"),
                "CPU_BASELINE_LIB": os.path.join(BUILD, "libcpu_baseline_asan.so")})
    return env


def test_golden_runner_over_the_oracle_under_asan_ubsan():
    env = _asan_env()
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-q", "-x",
                        "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "passed" in r.stdout and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def test_cpu_baseline_under_asan_ubsan():
    env = _asan_env()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "cpu_baseline_check.py")], capture_output=True, text=True,
                       timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "cpu_baseline OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def test_c_examples_compile_with_sanitizers():
    """The C hosts (examples/*.c) are strict C11 against the header; built with the sanitizers they still link and, with
    no device present, exit through their no-device path without a finding."""
    import torch

    libdir = os.path.join(ROOT, "arrow_gpu_amd", "lib")
    for name in ("simple", "arrow_cdata"):
        exe = os.path.join(ROOT, "tests", "cpp", "build", name + "_asan")
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        cmd = ["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-g", "-fsanitize=address,undefined",
               "-fno-sanitize-recover=all", os.path.join(ROOT, "examples", name + ".c"), "-I" + os.path.join(ROOT, "include"),
               f"-L{libdir}", "-larrow_gpu_hip", "-Wl,-rpath," + libdir, "-o", exe]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-3000:]
        if torch.cuda.device_count() == 0:
            env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
            r = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=env)
            assert r.returncode == 2, r.stdout[-1000:] + r.stderr[-2000:]


def test_ipc_reader_under_asan_ubsan_with_truncated_and_corrupted_input(tmp_path):
    """The Arrow IPC reader (csrc/arrow_ipc_reader.inc: plain C++, the host-only entry points) compiled by g++ with ASan +
    UBSan into tests/cpp/ipc_fuzz.cpp: every truncation on a grid and 2 × 1500 random byte flips of a pyarrow-written stream and
    file, each on a heap copy of exactly the mutated length, every byte a column view claims is read."""
    pytest.importorskip("pyarrow")
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_ipc_host import make_table, serialise

    exe = os.path.join(ROOT, "tests", "cpp", "build", "ipc_fuzz_asan")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    libdir = os.path.join(ROOT, "arrow_gpu_amd", "lib")
    src = os.path.join(ROOT, "tests", "cpp", "ipc_fuzz.cpp")
    deps = [src, os.path.join(ROOT, "arrow_gpu_amd", "csrc", "arrow_ipc_reader.inc"), os.path.join(ROOT, "include", "arrow_gpu.h")]
    if not (os.path.exists(exe) and all(os.path.getmtime(exe) >= os.path.getmtime(d) for d in deps)):
        # host C++ only (the reader has no HIP in it): g++, like the oracle's sanitizer build
        cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", src, "-o", exe,
               f"-L{libdir}", "-larrow_gpu_hip", "-Wl,-rpath," + libdir]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
    table = make_table(np.random.default_rng(1), 300)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    import pyarrow as pa

    cases = [("s.arrow", serialise(table, False, 100)), ("f.arrow", serialise(table, True, 100))]
    try:  # LZ4-compressed bodies: the frame decoder sees the same truncations and flips
        sink = pa.BufferOutputStream()
        with pa.ipc.new_file(sink, table.schema, options=pa.ipc.IpcWriteOptions(compression="lz4")) as w:
            for b in table.to_batches(max_chunksize=100):
                w.write_batch(b)
        cases.append(("lz4.arrow", sink.getvalue().to_pybytes()))
    except Exception:
        pass
    try:  # dictionary-encoded numeric columns: the index / dictionary decode sees the same mutations
        rngd = np.random.default_rng(3)
        dt = pa.table({"d": pa.DictionaryArray.from_arrays(pa.array(rngd.integers(0, 50, 400).astype(np.int16), mask=rngd.random(400) < 0.2),
                                                             pa.array(np.arange(50, dtype=np.float32) * 1.5)),
                       "p": pa.array(np.arange(400, dtype=np.int32))})
        cases.append(("dict.arrow", serialise(dt, True, 150)))
        cases.append(("dict_stream.arrow", serialise(dt, False, 150)))
    except Exception:
        pass
    # a hand-written schema whose struct Fields share ONE child table per level (k^depth walk) — byte flips alone
    # never reach that shape
    from test_ipc_host import shared_child_schema_stream

    cases.append(("shared_child.arrow", shared_child_schema_stream(10, 4)))
    cases.append(("shared_child_deep.arrow", shared_child_schema_stream(80, 4)))
    for name, blob in cases:
        path = tmp_path / name
        path.write_bytes(blob)
        r = subprocess.run([exe, str(path), "1500"], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0 and "ipc_fuzz OK" in r.stdout, name + ": " + r.stdout[-1000:] + r.stderr[-3000:]
        assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
