"""GPU: the multi-GPU half of the C ABI (include/arrow_gpu.h "multi-GPU") rehearsed on ONE GPU: an RCCL communicator
of world size 1 runs exactly the code a 2/4/8-GPU job runs (ncclCommInitRank, ncclAllGather of the 16-byte record,
rank-ordered combine kernel, ncclAllReduce), checked against the oracle's sharded-reduce spec.  The world-size-2
arithmetic of the combine (order, NaN rule, empty shards) is covered on CPU by tests/test_sharding_gloo.py against the
same spec; an 8-GPU node is the driver's to launch."""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from arrow_gpu_amd import _capi as capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice
    from arrow_gpu_amd.sharding import Communicator

    dev = GpuDevice(0)
    p = ArrowComputePipeline(dev, "comm")
    comm = Communicator.single(dev)
    yield dev, p, comm
    comm.close()


def bits(x, npd):
    return np.array([x], dtype=npd).view(np.uint32 if np.dtype(npd).itemsize == 4 else np.uint64)[0]


CASES = [(capi.F32, O.F32, np.float32), (capi.I32, O.I32, np.int32), (capi.U32, O.U32, np.uint32)]


@pytest.mark.parametrize("n", [0, 1, 255, 65536, 1_000_003, 16_777_216 + 5])
def test_comm_reduce_world1_matches_the_sharded_spec(ctx, n):
    dev, p, comm = ctx
    out = dev.create_empty_buffer(16)
    for dt, odt, npd in CASES:
        host = (O.synth_f32(n, 9, 0, -1.0, 1.0) if dt == capi.F32 else O.synth_i32(n, 9, 0, 0).view(npd))
        buf = dev.create_gpu_buffer_with_data(host) if n else dev.create_empty_buffer(16)
        for op in (capi.RED_SUM, capi.RED_MIN, capi.RED_MAX):
            comm.reduce(p, op, dt, buf, None, n, out)
            got = dev.retrive_data(out, 4, pipeline=p).view(npd)[0]
            exp = O.sharded_reduce(op, odt, [host])
            assert bits(got, npd) == bits(exp, npd), (dt, op, n, got, exp)
    if n:
        host = O.synth_f32(n, 9, 0, -1.0, 1.0)
        buf = dev.create_gpu_buffer_with_data(host)
        comm.reduce_sum_f64(p, buf, None, n, out)
        got = dev.retrive_data(out, 8, pipeline=p).view(np.float64)[0]
        assert got == O.sharded_reduce_sum_f64([host])


def test_comm_reduce_with_validity_and_nan(ctx):
    dev, p, comm = ctx
    n = 300_001
    host = O.synth_f32(n, 3, 0, -5.0, 5.0)
    host[::7] = np.nan
    v = O.synth_bits(n, 4, 0, 0.7)
    buf, dv, out = dev.create_gpu_buffer_with_data(host), dev.create_gpu_buffer_with_data(v), dev.create_empty_buffer(16)
    for op in (capi.RED_MIN, capi.RED_MAX, capi.RED_SUM):
        comm.reduce(p, op, capi.F32, buf, dv, n, out)
        got = dev.retrive_data(out, 4, pipeline=p).view(np.float32)[0]
        exp = O.sharded_reduce(op, O.F32, [host], [v])
        assert bits(got, np.float32) == bits(exp, np.float32) or (np.isnan(got) and np.isnan(exp)), (op, got, exp)
    allnan = np.full(1000, np.nan, np.float32)
    b2 = dev.create_gpu_buffer_with_data(allnan)
    comm.reduce(p, capi.RED_MIN, capi.F32, b2, None, 1000, out)
    assert np.isnan(dev.retrive_data(out, 4, pipeline=p).view(np.float32)[0])


def test_comm_final_reduce_all_reduce_and_barrier(ctx):
    dev, p, comm = ctx
    part = dev.create_gpu_buffer_with_data(np.array([12345], np.uint32))
    out = dev.create_empty_buffer(16)
    comm.final_reduce(p, capi.RED_SUM, capi.U32, part, 10, out)
    assert dev.retrive_data(out, 4, pipeline=p).view(np.uint32)[0] == 12345
    comm.final_reduce(p, capi.RED_MAX, capi.U32, part, 0, out)  # an empty shard contributes the identity
    assert dev.retrive_data(out, 4, pipeline=p).view(np.uint32)[0] == 0
    pf = dev.create_gpu_buffer_with_data(np.array([1.5], np.float64))
    comm.final_reduce(p, capi.RED_SUM, capi.F32, pf, 3, out, f64=True)
    assert dev.retrive_data(out, 8, pipeline=p).view(np.float64)[0] == 1.5
    cnt = dev.create_gpu_buffer_with_data(np.array([7, 9], np.uint64))
    comm.all_reduce(p, capi.RED_SUM, capi.COMM_U64, cnt, 2)
    assert dev.retrive_data(cnt, 16, pipeline=p).view(np.uint64).tolist() == [7, 9]
    comm.barrier(p)
    r, w = C.c_int32(-1), C.c_int32(-1)
    capi.call("agpu_comm_rank", comm._h, C.byref(r), C.byref(w))
    assert (r.value, w.value) == (0, 1)


def test_sharded_sum_of_256k_row_shards_is_the_reference_tree(ctx):
    """Property the final reduce is built for: cut a column into shards of 256^k rows, reduce each with the reference's
    tree, combine the shard sums with one more adjacent-pair level → bit-identical to the reference's tree over the
    whole column.  Checked here with the GPU doing the per-shard part and the oracle the whole-column tree."""
    dev, p, comm = ctx
    k_rows = 65536
    shards = 8
    host = O.synth_f32(k_rows * shards, 21, 0, -1000.0, 1000.0)
    whole = O.reduce(O.RED_SUM, O.F32, host)
    parts = np.empty(shards, np.float32)
    out = dev.create_empty_buffer(16)
    buf = dev.create_gpu_buffer_with_data(host)
    for s in range(shards):
        capi.call("agpu_reduce", p._handle, capi.RED_SUM, capi.F32, C.c_void_p(buf.ptr + 4 * k_rows * s), None, k_rows, C.c_void_p(out.ptr))
        parts[s] = dev.retrive_data(out, 4, pipeline=p).view(np.float32)[0]
    combined = O.reduce(O.RED_SUM, O.F32, parts)  # what comm_finish_sum_f32_kernel does with the gathered records
    assert bits(combined, np.float32) == bits(whole, np.float32)
    assert bits(O.sharded_reduce(O.RED_SUM, O.F32, np.split(host, shards)), np.float32) == bits(whole, np.float32)


@pytest.mark.parametrize("world", [2, 3, 8, 255, 256])
def test_combine_of_many_ranks_records_matches_the_spec(ctx, world):
    """The rank-ordered combine kernels with world > 1 — the step a 1-GPU box cannot reach through RCCL — fed with
    hand-made records (NaN shards, empty shards, wrapping integer sums) and checked against oracle.combine_records."""
    dev, p, _ = ctx
    rng = np.random.default_rng(world)
    out = dev.create_empty_buffer(16)
    for dt, odt, npd in CASES:
        if npd is np.float32:
            vals = (rng.standard_normal(world) * 1e3).astype(np.float32)
            vals[rng.random(world) < 0.2] = np.nan
        else:
            vals = rng.integers(np.iinfo(npd).min, int(np.iinfo(npd).max) + 1, world, dtype=np.int64).astype(npd)
        n_local = rng.integers(0, 3, world).astype(np.uint64)  # a third of the shards are empty
        for op in (capi.RED_SUM, capi.RED_MIN, capi.RED_MAX):
            v = vals.copy()
            if op == capi.RED_SUM:
                v[n_local == 0] = 0  # an empty shard's local sum is 0
                if npd is np.float32:
                    v = np.nan_to_num(v, nan=1.5)
            rec = np.zeros((world, 2), np.uint64)
            rec[:, 0] = v.view(np.uint32).astype(np.uint64)
            rec[:, 1] = n_local
            drec = dev.create_gpu_buffer_with_data(rec)
            capi.call("agpu_reduce_combine", p._handle, op, dt, 0, C.c_void_p(drec.ptr), world, C.c_void_p(out.ptr))
            got = dev.retrive_data(out, 4, pipeline=p).view(npd)[0]
            exp = O.combine_records(op, odt, [(v[r], int(n_local[r])) for r in range(world)])
            assert bits(got, npd) == bits(exp, npd) or (npd is np.float32 and np.isnan(got) and np.isnan(exp)), (dt, op, got, exp)
    sums = rng.standard_normal(world)
    rec = np.zeros((world, 2), np.uint64)
    rec[:, 0] = sums.view(np.uint64)
    rec[:, 1] = 1
    drec = dev.create_gpu_buffer_with_data(rec)
    capi.call("agpu_reduce_combine", p._handle, capi.RED_SUM, capi.F32, 1, C.c_void_p(drec.ptr), world, C.c_void_p(out.ptr))
    acc = 0.0
    for x in sums:
        acc = acc + float(x)
    assert dev.retrive_data(out, 8, pipeline=p).view(np.float64)[0] == acc


# ---- round 3: first-contact proofing of the multi-rank path
_TIMEOUT_WORKER = r"""
import os, sys, time
sys.path.insert(0, {root!r})
from arrow_gpu_amd._capi import ArrowErrorGPU
from arrow_gpu_amd.gpu_utils import GpuDevice
from arrow_gpu_amd.sharding import Communicator
dev = GpuDevice(0)
t0 = time.monotonic()
try:
    Communicator(dev, 0, 2, Communicator.unique_id(), timeout_s=3.0)   # rank 1 of 2 never arrives
except ArrowErrorGPU as e:
    dt = time.monotonic() - t0
    sys.stdout.write("TIMEOUT %.2f %s\n" % (dt, e))
    sys.stdout.flush()
    os._exit(0 if 2.5 < dt < 20 else 3)   # the pending RCCL rendezvous thread cannot be joined: end the process
os._exit(4)
"""


def test_comm_init_gives_up_when_a_rank_never_arrives():
    """agpu_comm_init_rank_timeout: rank 0 of a world of 2 alone → AGPU_ERR_HIP after the deadline, not a hang (a fresh
    process: the pending ncclCommInitRank stays behind on its helper thread)"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _TIMEOUT_WORKER.format(root=root)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert "TIMEOUT" in r.stdout and "gave up after 3000 ms" in r.stdout


def test_comm_shared_by_two_pipelines_is_ordered(ctx):
    """the record buffers of a communicator are shared: a call made on ANOTHER pipeline is ordered behind the previous one
    (two streams raced on c->send / c->recv)"""
    from arrow_gpu_amd.gpu_utils import ArrowComputePipeline

    dev, p, comm = ctx
    q = ArrowComputePipeline(dev, "comm-2")
    n = 40_000_003
    a, b = O.synth_f32(n, 21, 0, -1.0, 1.0), O.synth_f32(n, 22, 0, -1.0, 1.0)
    da, db = dev.create_gpu_buffer_with_data(a), dev.create_gpu_buffer_with_data(b)
    dev.sync()
    outs = [dev.create_empty_buffer(16) for _ in range(8)]
    for k in range(4):  # alternate streams without any host sync in between
        comm.reduce(p, capi.RED_SUM, capi.F32, da, None, n, outs[2 * k])
        comm.reduce(q, capi.RED_MAX, capi.F32, db, None, n, outs[2 * k + 1])
    p.sync(), q.sync()
    es, em = O.sharded_reduce(O.RED_SUM, O.F32, [a]), O.sharded_reduce(O.RED_MAX, O.F32, [b])
    for k in range(4):
        assert bits(dev.retrive_data(outs[2 * k], 4).view(np.float32)[0], np.float32) == bits(es, np.float32)
        assert bits(dev.retrive_data(outs[2 * k + 1], 4).view(np.float32)[0], np.float32) == bits(em, np.float32)


def test_comm_from_env_and_runtime_info(monkeypatch, tmp_path):
    """the bench worker's way in: ranks from the environment, id over the file rendezvous — at world 1 here; and the
    process reports which librccl / libamdhip64 it runs on (one runtime: no torch in this process's product path)"""
    from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice
    from arrow_gpu_amd.sharding import Communicator

    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.setenv(k, {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"}[k])
    dev = GpuDevice(0)
    p = ArrowComputePipeline(dev, "env")
    comm = Communicator.from_env(dev)
    assert (comm.rank, comm.world) == (0, 1)
    comm.barrier(p)
    comm.close()
    # the file path itself, world 1 (rank 0 is also the only reader)
    comm = Communicator.from_file(dev, 0, 1, str(tmp_path / "id"), timeout_s=10.0)
    comm.barrier(p)
    comm.close()
    assert list(tmp_path.iterdir()) == []
    info = Communicator.runtime_info()
    assert "rccl" in info and "librccl" in info and "libamdhip64" in info


# ---- round 4: a multi-GPU record that proves itself and a failed collective that cannot hang the host
def test_comm_size_and_peers_report_what_rccl_sees(ctx):
    """agpu_comm_size = ncclCommCount / UserRank / CuDevice; agpu_comm_peers gathers one identity record per rank THROUGH the
    communicator — at world 1: this process, this GPU; `sharding.world_proof` turns the records into the n_gpus a line may claim"""
    import os

    from arrow_gpu_amd import sharding

    dev, p, comm = ctx
    n, r, d = comm.size()
    assert (n, r) == (1, 0) and d == dev.ordinal
    peers = comm.peers(p)
    assert len(peers) == 1
    me = peers[0]
    assert me["rank"] == 0 and me["world"] == 1 and me["pid"] == os.getpid() and me["device_ordinal"] == dev.ordinal
    assert me["arch"].startswith("gfx950") and len(me["uuid"]) == 32 and len(me["host"]) == 16
    ident = sharding.Peer()
    capi.call("agpu_device_identity", dev._handle, C.byref(ident))
    same = ident.as_dict()
    assert (same["pci"], same["uuid"], same["host"]) == (me["pci"], me["uuid"], me["host"]) and same["rank"] == -1
    proof = sharding.world_proof(peers, 1)
    assert proof["ok"] and proof["rccl_ranks"] == 1 and proof["distinct_devices"] == 1
    assert not sharding.world_proof(peers, 8)["ok"]       # --gpus 8 on a one-rank communicator is refused
    assert not sharding.world_proof(peers + peers, 2)["ok"]  # two records of ONE device are not two GPUs
    # the deadline-aware wait behind a collective
    out = dev.create_empty_buffer(16)
    buf = dev.create_gpu_buffer_with_data(O.synth_f32(100_000, 1, 0, -1.0, 1.0))
    comm.reduce(p, capi.RED_MAX, capi.F32, buf, None, 100_000, out)
    comm.sync(p)
    assert dev.retrive_data(out, 4, pipeline=p).view(np.float32)[0] == O.synth_f32(100_000, 1, 0, -1.0, 1.0).max()


_POISON_WORKER = r"""
import os, sys, time
sys.path.insert(0, {root!r})
import numpy as np
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd._capi import ArrowErrorGPU
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice
from arrow_gpu_amd.sharding import Communicator
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "poison")
buf = dev.create_gpu_buffer_with_data(np.arange(1000, dtype=np.float32))
try:
    Communicator(dev, 0, 2, Communicator.unique_id(), timeout_s=2.0)   # rank 1 of 2 never arrives → the device is poisoned
    os._exit(10)
except ArrowErrorGPU:
    pass
t0 = time.monotonic()
fails = 0
for call in (lambda: p.sync(), lambda: dev.sync(), lambda: dev.create_empty_buffer(1 << 20), lambda: dev.retrive_data(buf, 4000, pipeline=p)):
    try:
        call()
    except ArrowErrorGPU as e:
        fails += 1 if "poisoned" in str(e) else 0
# destructors / destroy calls return instead of waiting for the device
del buf
capi.lib().agpu_pipeline_destroy(p._handle); p._handle = None
capi.lib().agpu_device_destroy(dev._handle); dev._handle = None
dt = time.monotonic() - t0
sys.stdout.write("POISONED fails=%d dt=%.2f\n" % (fails, dt))
sys.stdout.flush()
os._exit(0 if fails == 4 and dt < 10 else 5)
"""


def test_a_timed_out_collective_poisons_the_device_and_nothing_hangs_afterwards():
    """after a collective deadline every call fails fast and the destroy calls leak instead of synchronising, so a host
    unwinding through destructors ends (a fresh process: the pending RCCL call stays behind on its helper thread)"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _POISON_WORKER.format(root=root)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert "POISONED fails=4" in r.stdout


# ---- round 5: a ONE-rank communicator never dies of RCCL's socket bootstrap
_LOCAL_WORKER = r"""
import os, sys, ctypes as C
sys.path.insert(0, {root!r})
import numpy as np
import oracle as O
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice
from arrow_gpu_amd.sharding import Communicator, world_proof
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "local")
comm = Communicator(dev, 0, 1, Communicator.unique_id(), timeout_s=0.5)   # the helper sits in its stall: the deadline passes
assert comm.is_local, "expected the local fallback"
assert comm.size() == (1, 0, 0)
n = 1 << 20
x = O.synth_f32(n, 11, 0, -1.0, 1.0)
d = dev.create_gpu_buffer_with_data(x)
out = dev.create_empty_buffer(16)
for op, red in ((capi.RED_SUM, O.RED_SUM), (capi.RED_MIN, O.RED_MIN), (capi.RED_MAX, O.RED_MAX)):
    comm.reduce(p, op, capi.F32, d, None, n, out)
    comm.sync(p)
    got = dev.retrive_data(out, 4, pipeline=p).view(np.float32)[0]
    assert np.float32(got).view(np.uint32) == np.float32(O.sharded_reduce(red, O.F32, [x])).view(np.uint32), op
cnt = dev.create_gpu_buffer_with_data(np.array([7, 9], np.uint64))
comm.all_reduce(p, capi.RED_SUM, capi.COMM_U64, cnt, 2)
comm.barrier(p)
assert dev.retrive_data(cnt, 16, pipeline=p).view(np.uint64).tolist() == [7, 9]
peers = comm.peers(p)
proof = world_proof(peers, 1)
assert proof["ok"] and proof["rccl_ranks"] == 1 and proof["distinct_devices"] == 1, proof
comm.close()
# and the device is NOT poisoned: ordinary work and a second, real communicator still run
real = Communicator.single(dev)
print("second communicator local:", real.is_local)
real.barrier(p)
real.close()
print("LOCAL-OK")
sys.stdout.flush()
os._exit(0)   # the parked helper thread cannot be joined
"""


def test_a_one_rank_communicator_survives_a_bootstrap_that_does_not_come_up():
    """seen once in round 5: ncclCommInitRank of a ONE-rank communicator pending for 60 s on a shared node.  With the helper thread stalled
    (AGPU_COMM_TEST_STALL_INIT_MS) the init deadline passes and the communicator is created LOCAL: reductions equal the sharded spec, the
    identity record still proves one device, nothing is poisoned"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AGPU_COMM_TEST_STALL_INIT_MS="4000", AGPU_LIB=os.path.join(root, "arrow_gpu_amd", "lib", "libarrow_gpu_hip_hooks.so"))
    r = subprocess.run([sys.executable, "-c", _LOCAL_WORKER.format(root=root)], capture_output=True, text=True, timeout=180, env=env)
    assert r.returncode == 0 and "LOCAL-OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])


_LATE_WORKER = r"""
import os, sys, time
sys.path.insert(0, {root!r})
import numpy as np
from arrow_gpu_amd import _capi as capi
from arrow_gpu_amd.gpu_utils import ArrowComputePipeline, GpuDevice
from arrow_gpu_amd.sharding import Communicator
dev = GpuDevice(0)
p = ArrowComputePipeline(dev, "late")
comm = Communicator(dev, 0, 1, Communicator.unique_id(), timeout_s=0.5)   # the helper calls RCCL 0.3 s in and returns after the deadline
assert comm.is_local, "expected the local fallback"
cnt = dev.create_gpu_buffer_with_data(np.array([7, 9], np.uint64))
comm.all_reduce(p, capi.RED_SUM, capi.COMM_U64, cnt, 2)
comm.barrier(p)
time.sleep({sleep})    # the late ncclComm_t arrives meanwhile (or, with sleep 0, is still on its way when close() looks)
comm.close()
print("LATE-OK")
sys.stdout.flush()
# a NORMAL interpreter exit: with RCCL's communicator left alive this is where the process died
"""


@pytest.mark.parametrize("sleep", [6, 0])
def test_a_bootstrap_that_comes_up_late_is_cleaned_up(sleep):
    """AGPU_COMM_TEST_STALL_INIT_MS=19000 against the 20 s one-rank deadline, examples/sharded_stats: the helper's ncclCommInitRank returned a
    second after the call had fallen back to the local communicator, and the process died with SIGSEGV at exit (RCCL's communicator and its
    proxy threads still alive).  agpu_comm_destroy now takes the late communicator down (or tells the helper to)"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AGPU_COMM_TEST_STALL_INIT_MS="300", AGPU_LIB=os.path.join(root, "arrow_gpu_amd", "lib", "libarrow_gpu_hip_hooks.so"))
    r = subprocess.run([sys.executable, "-c", _LATE_WORKER.format(root=root, sleep=sleep)], capture_output=True, text=True, timeout=180, env=env)
    assert r.returncode == 0 and "LATE-OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
