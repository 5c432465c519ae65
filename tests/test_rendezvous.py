"""CPU: the torch-free launch path of the multi-GPU bench — ranks from the launcher's environment and the file rendezvous
that ships the RCCL id (arrow_gpu_amd/sharding.py) — with world_size 2 and 4 as SEPARATE PROCESSES, without torch.
The RCCL half (agpu_comm_init_rank with its deadline) needs a GPU: tests/test_gpu_comm.py."""
import os
import subprocess
import sys
import time

import pytest

from arrow_gpu_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import os, sys, time
sys.path.insert(0, {root!r})
from arrow_gpu_amd import sharding
assert "torch" not in sys.modules
rank, world, local = sharding.ranks_from_env()
time.sleep(float(os.environ.get("DELAY", "0")))
path = sharding.rendezvous_path_from_env()
payload = sharding.file_rendezvous(path, rank, world, (lambda: bytes([7]) * 128) if rank == 0 else None, float(os.environ.get("TMO", "20")))
assert "torch" not in sys.modules
sys.stdout.write("%d %d %d %s %s\n" % (rank, world, local, payload.hex()[:8], path))
sharding.file_rendezvous_cleanup(path, rank)
"""


def _spawn(rank, world, env_extra):
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_PORT="29999", **env_extra)
    return subprocess.Popen([sys.executable, "-c", _WORKER.format(root=ROOT)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


@pytest.mark.parametrize("world", [2, 4])
def test_file_rendezvous_between_processes_without_torch(tmp_path, world):
    path = str(tmp_path / "rdzv")
    procs = [_spawn(r, world, {"AGPU_RENDEZVOUS_FILE": path, "DELAY": str(0.1 * ((r * 3) % world))}) for r in range(world)]
    outs = [p.communicate(timeout=60) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    seen = sorted(o[0].split()[0] for o in outs)
    assert seen == [str(r) for r in range(world)]
    assert all(o[0].split()[3] == "07070707" for o in outs)
    assert os.listdir(tmp_path) == []  # every rank removed its files


def test_rendezvous_ignores_files_of_an_older_run_at_the_same_path(tmp_path):
    path = str(tmp_path / "rdzv")
    for r in range(2):
        (tmp_path / f"rdzv.ready.{r}").write_bytes(b"o" * 16)
        (tmp_path / f"rdzv.ack.{r}").write_bytes(b"o" * 16)
    (tmp_path / "rdzv.id").write_bytes(b"S" * (128 + 32))  # a stale id: must never be handed out
    for delays in (("0", "0.3"), ("0.3", "0")):
        procs = [_spawn(r, 2, {"AGPU_RENDEZVOUS_FILE": path, "DELAY": delays[r]}) for r in range(2)]
        outs = [p.communicate(timeout=60) for p in procs]
        assert all(p.returncode == 0 for p in procs), outs
        assert all(o[0].split()[3] == "07070707" for o in outs)


def test_a_rank_that_never_arrives_is_a_timeout_not_a_hang(tmp_path):
    path = str(tmp_path / "rdzv")
    t0 = time.monotonic()
    p0 = _spawn(0, 2, {"AGPU_RENDEZVOUS_FILE": path, "TMO": "1.0"})
    out = p0.communicate(timeout=60)
    assert p0.returncode != 0 and "TimeoutError" in out[1] and "[1]" in out[1]
    p1 = _spawn(1, 2, {"AGPU_RENDEZVOUS_FILE": path + "b", "TMO": "1.0"})
    out = p1.communicate(timeout=60)
    assert p1.returncode != 0 and "TimeoutError" in out[1]
    assert time.monotonic() - t0 < 30


def test_default_path_is_shared_by_the_ranks_of_a_launch():
    # under torchrun (MASTER_PORT exported): the path depends on address / port / run id / restart count only — wrapper
    # processes between launcher and worker do not matter; another port or run id → another path
    procs = [_spawn(r, 2, {}) for r in range(2)]
    outs = [p.communicate(timeout=60) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    paths = {o[0].split()[4] for o in outs}
    assert len(paths) == 1 and "29999" in paths.pop()
    a = sharding.rendezvous_path_from_env({"MASTER_PORT": "1"})
    b = sharding.rendezvous_path_from_env({"MASTER_PORT": "2"})
    c = sharding.rendezvous_path_from_env({"MASTER_PORT": "1", "TORCHELASTIC_RUN_ID": "x"})
    assert len({a, b, c}) == 3
    assert sharding.rendezvous_path_from_env({"AGPU_RENDEZVOUS_FILE": "/x/y"}) == "/x/y"
    # no MASTER_PORT (mpirun / srun): siblings of one launcher share the parent's pid + start time
    assert f"ppid{os.getppid()}_" in sharding.rendezvous_path_from_env({"OMPI_COMM_WORLD_RANK": "0"})
    # a second launch on the same port right after the first (files possibly left behind) still works: nonces
    for _ in range(2):
        procs = [_spawn(r, 2, {"DELAY": str(0.2 * r)}) for r in range(2)]
        outs = [p.communicate(timeout=60) for p in procs]
        assert all(p.returncode == 0 for p in procs), outs


def test_ranks_from_env_knows_the_common_launchers():
    assert sharding.ranks_from_env({}) == (0, 1, 0)
    assert sharding.ranks_from_env({"RANK": "3", "WORLD_SIZE": "8", "LOCAL_RANK": "3"}) == (3, 8, 3)
    assert sharding.ranks_from_env({"OMPI_COMM_WORLD_RANK": "1", "OMPI_COMM_WORLD_SIZE": "2", "OMPI_COMM_WORLD_LOCAL_RANK": "1"}) == (1, 2, 1)
    assert sharding.ranks_from_env({"SLURM_PROCID": "5", "SLURM_NTASKS": "8", "SLURM_LOCALID": "1"}) == (5, 8, 1)
    with pytest.raises(ValueError):
        sharding.ranks_from_env({"RANK": "2", "WORLD_SIZE": "2"})


def test_bench_worker_never_imports_torch():
    """bench.py's product path: one HIP + RCCL runtime per process — no torch import anywhere in it"""
    import re

    stmt = re.compile(r"^\s*(import torch|from torch)", re.M)
    assert not stmt.search(open(os.path.join(ROOT, "bench.py")).read())
    for name in ("_capi.py", "gpu_utils.py", "array.py"):
        assert not stmt.search(open(os.path.join(ROOT, "arrow_gpu_amd", name)).read())


def test_bench_started_plainly_with_gpus_n_launches_its_own_workers():
    """`python bench.py --gpus 2` without a launcher: the parent starts one child per GPU (torchrun's environment) and never
    touches a device itself.  Without a GPU both children fail at device creation and the parent reports it — no hang, no line."""
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "OMPI_COMM_WORLD_RANK", "SLURM_PROCID")}
    env["HIP_VISIBLE_DEVICES"] = "-1"  # also on a GPU box: this test is about the plumbing
    env["ROCR_VISIBLE_DEVICES"] = "-1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--rendezvous-timeout", "5"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1
    assert r.stdout.strip() == ""
    assert "worker exit codes [1, 1]" in r.stderr and "NoDevice" in r.stderr


def test_loopback_bootstrap_is_named_only_when_every_rank_is_on_this_host():
    """sharding.prefer_loopback_bootstrap: RCCL's rendezvous socket goes over `lo` for a one-rank communicator and for a launcher whose
    MASTER_ADDR is the loopback address; a caller's own choice and a multi-node address are left alone"""
    from arrow_gpu_amd.sharding import prefer_loopback_bootstrap

    e = {}
    assert prefer_loopback_bootstrap(1, e) and e["NCCL_SOCKET_IFNAME"] == "lo"
    e = {"MASTER_ADDR": "127.0.0.1"}
    assert prefer_loopback_bootstrap(8, e) and e["NCCL_SOCKET_IFNAME"] == "lo"
    e = {"MASTER_ADDR": "10.0.0.7"}
    assert not prefer_loopback_bootstrap(8, e) and "NCCL_SOCKET_IFNAME" not in e
    e = {}
    assert not prefer_loopback_bootstrap(8, e) and "NCCL_SOCKET_IFNAME" not in e
    e = {"MASTER_ADDR": "127.0.0.1", "NCCL_SOCKET_IFNAME": "eth0"}
    assert not prefer_loopback_bootstrap(8, e) and e["NCCL_SOCKET_IFNAME"] == "eth0"
