"""Chunk-sharding of a column across the GPUs of one node + the final reduce of whole-column statistics.

Not in the reference (it is single-device, single-queue: SURVEY §5); this is north_star's multi-GPU rule:
one process per GPU, every rank owns ONE contiguous row range of the column, element-wise / compare / cast / bitmap
kernels need no communication (outputs stay sharded), and only sum / min / max / popcount finish with a collective of
ONE 16-byte record per statistic over the C ABI's RCCL communicator (`Communicator`: rendezvous over a file with a
deadline, over torch.distributed when the caller already has a group, or in-process between threads).  The message is
tiny, so the collective is latency-bound and independent of the xGMI link rate.

Shard boundaries fall on multiples of 512 rows so every rank owns whole 64-bit bitmap words and 2 KiB-aligned f32
spans (16-byte vector path, no split validity words).
"""
from __future__ import annotations

import contextlib
import ctypes as C
from dataclasses import dataclass

from . import _capi as capi

ROW_ALIGN = 512


@dataclass(frozen=True)
class Shard:
    rank: int
    world: int
    row0: int   # first row of this rank's range
    rows: int   # number of rows in the range

    @property
    def row_end(self) -> int:
        return self.row0 + self.rows


def shard_rows(total_rows: int, world: int, rank: int, align: int = ROW_ALIGN) -> Shard:
    """Contiguous, near-equal ranges; every boundary except the column end is a multiple of `align` rows."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank {rank} for world {world}")
    if total_rows < 0:
        raise ValueError("negative row count")
    chunks = (total_rows + align - 1) // align
    per, extra = divmod(chunks, world)
    first_chunk = rank * per + min(rank, extra)
    n_chunks = per + (1 if rank < extra else 0)
    row0 = min(first_chunk * align, total_rows)
    row_end = min((first_chunk + n_chunks) * align, total_rows)
    return Shard(rank, world, row0, row_end - row0)


def all_shards(total_rows: int, world: int, align: int = ROW_ALIGN):
    return [shard_rows(total_rows, world, r, align) for r in range(world)]


def shard_batches(batch_rows, world: int, rank: int):
    """Record batches of an Arrow IPC file are the natural shard unit (each is self-contained, with whole bitmap words):
    rank r gets a CONTIGUOUS run of batches, so that rank order = row order — what the rank-ordered final reduce
    (`Communicator.reduce`) assumes.  Runs are cut where the cumulative row count crosses r/world of the total, so the
    ranks' row counts differ by at most one batch.  Returns range(first, last)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank {rank} for world {world}")
    total = sum(batch_rows)
    cuts, acc, b = [0], 0, 0
    for r in range(1, world):
        target = total * r / world
        while b < len(batch_rows) and acc + batch_rows[b] / 2 <= target:
            acc += batch_rows[b]
            b += 1
        cuts.append(b)
    cuts.append(len(batch_rows))
    return range(cuts[rank], cuts[rank + 1])


def read_ipc_shard(source, device, rank: int, world: int, columns=None) -> dict:
    """This rank's share of an Arrow IPC file / stream: {column: [GPU array per record batch]} for the contiguous run of
    batches `shard_batches` assigns to `rank` (every rank maps the same file; only its own batches are uploaded)."""
    from .ipc import IpcReader

    out: dict = {}
    with IpcReader(source) as r:
        rows = [r.batch_rows(b) for b in range(r.num_batches)]
        for b in shard_batches(rows, world, rank):
            for name, arr in r.read_batch(b, device, columns).items():
                out.setdefault(name, []).append(arr)
    return out


def ranks_from_env(env=None):
    """(rank, world, local_rank) from the launcher's environment — torch.distributed.run's names first, then Open MPI's
    and Slurm's; (0, 1, 0) when none is set."""
    import os

    e = os.environ if env is None else env
    for r, w, l in (("RANK", "WORLD_SIZE", "LOCAL_RANK"), ("OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_RANK"),
                    ("SLURM_PROCID", "SLURM_NTASKS", "SLURM_LOCALID")):
        if r in e and w in e:
            rank, world = int(e[r]), int(e[w])
            if world < 1 or not (0 <= rank < world):
                raise ValueError(f"bad {r}={rank} / {w}={world}")
            return rank, world, int(e.get(l, rank))
    return 0, 1, 0


def rendezvous_path_from_env(env=None) -> str:
    """A path that is the same for all ranks of ONE launch: AGPU_RENDEZVOUS_FILE if set; under torch.distributed.run (or anything
    else that exports MASTER_PORT) <tmp>/agpu_rdzv_<MASTER_ADDR>_<MASTER_PORT>_<run id>_<restart count> — two launches cannot hold
    one port at a time, and files an EARLIER launch left at the same path are harmless (`file_rendezvous` only accepts an id that
    carries this process's fresh nonce), so nothing about the process tree is assumed (wrapper scripts between launcher and worker
    are fine); without MASTER_PORT (mpirun, srun) the parent's pid and start time tell launches apart — there the workers must be
    siblings."""
    import os
    import tempfile

    e = os.environ if env is None else env
    if e.get("AGPU_RENDEZVOUS_FILE"):
        return e["AGPU_RENDEZVOUS_FILE"]
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    if e.get("MASTER_PORT"):
        tag = "_".join(str(x).replace("/", "-") for x in (e.get("MASTER_ADDR", "local"), e["MASTER_PORT"], e.get("TORCHELASTIC_RUN_ID", "none"),
                                                          e.get("TORCHELASTIC_RESTART_COUNT", "0")))
        return os.path.join(base, "agpu_rdzv_" + tag)
    ppid = os.getppid()
    start = "0"
    try:
        with open(f"/proc/{ppid}/stat") as f:
            start = f.read().rsplit(")", 1)[1].split()[19]  # field 22: start time in clock ticks since boot
    except (OSError, IndexError):
        pass
    return os.path.join(base, f"agpu_rdzv_ppid{ppid}_{start}")


def file_rendezvous(path: str, rank: int, world: int, make_payload=None, timeout_s: float = 60.0, payload_bytes: int = capi.COMM_ID_BYTES) -> bytes:
    """Ship `payload_bytes` bytes from rank 0 to every rank through the filesystem, with a deadline and proof against files
    left behind by an earlier run at the same path:
      every rank r  writes  <path>.ready.<r> = a fresh 16-byte nonce                         (atomically: tmp + rename)
      rank 0        waits for all `world` ready files, writes <path>.id = payload ‖ nonce_0 ‖ … ‖ nonce_{world-1}, and keeps
                    re-reading the ready files — a nonce that changes (it had picked up an older run's file before this
                    launch's rank replaced it) re-issues the id — until every rank has acknowledged
      rank r        waits for an id file that carries ITS nonce at position r (a stale file cannot), writes
                    <path>.ack.<r> = its nonce, and returns the payload.
    A rank that never arrives makes the others raise TimeoutError after `timeout_s` instead of blocking in RCCL.
    `make_payload` is called on rank 0 only, once."""
    import os
    import time

    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank {rank} for world {world}")
    nonce = os.urandom(16)

    def put(name, data):
        tmp = f"{name}.tmp.{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, name)

    def get(name, size):
        try:
            with open(name, "rb") as f:
                data = f.read()
            return data if len(data) == size else None
        except OSError:
            return None

    deadline = time.monotonic() + timeout_s
    # acks and ids of an older run at this path: an ack can only be fresh AFTER rank 0 has issued an id, so rank 0 may
    # safely clear them all before it starts
    for stale in ([f"{path}.id"] + [f"{path}.ack.{r}" for r in range(world)] if rank == 0 else [f"{path}.ack.{rank}"]):
        try:
            os.unlink(stale)
        except OSError:
            pass
    put(f"{path}.ready.{rank}", nonce)
    id_size = payload_bytes + 16 * world
    if rank != 0:
        while True:
            data = get(f"{path}.id", id_size)
            if data is not None and data[payload_bytes + 16 * rank: payload_bytes + 16 * (rank + 1)] == nonce:
                put(f"{path}.ack.{rank}", nonce)
                return data[:payload_bytes]
            if time.monotonic() > deadline:
                raise TimeoutError(f"rendezvous at {path}: rank {rank} saw no id from rank 0 within {timeout_s} s")
            time.sleep(0.005)
    payload, issued = None, None
    while True:
        nonces = [get(f"{path}.ready.{r}", 16) for r in range(world)]
        nonces[0] = nonce
        if all(x is not None for x in nonces):
            if payload is None:
                payload = make_payload()
                if len(payload) != payload_bytes:
                    raise ValueError("payload size")
            if nonces != issued:
                put(f"{path}.id", payload + b"".join(nonces))
                issued = nonces
            if all(get(f"{path}.ack.{r}", 16) == issued[r] for r in range(1, world)):
                return payload
        if time.monotonic() > deadline:
            missing = [r for r in range(world) if nonces[r] is None] or \
                      [r for r in range(1, world) if get(f"{path}.ack.{r}", 16) != (issued or nonces)[r]]
            raise TimeoutError(f"rendezvous at {path}: ranks {missing} of {world} never arrived within {timeout_s} s")
        time.sleep(0.005)


def file_rendezvous_cleanup(path: str, rank: int, wait_s: float = 5.0) -> None:
    """Remove this rank's files.  Rank 0 returns from `file_rendezvous` only after every ack, so it may delete the id at
    once; the others keep their ack until the id has gone (rank 0 may not have read the ack yet), for at most `wait_s`."""
    import os
    import time

    if rank != 0:
        t0 = time.monotonic()
        while os.path.exists(f"{path}.id") and time.monotonic() - t0 < wait_s:
            time.sleep(0.002)
    for name in ([f"{path}.ready.{rank}", f"{path}.ack.{rank}"] + ([f"{path}.id"] if rank == 0 else [])):
        try:
            os.unlink(name)
        except OSError:
            pass


class Peer(C.Structure):
    """include/arrow_gpu.h `agpu_comm_peer`: one rank's identity as gathered through the communicator."""
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("device_ordinal", C.c_int32), ("nccl_device", C.c_int32),
                ("pci_domain", C.c_int32), ("pci_bus", C.c_int32), ("pci_device", C.c_int32), ("pid", C.c_int32),
                ("host_hash", C.c_uint64), ("uuid", C.c_uint8 * 16), ("gcn_arch", C.c_char * 16)]

    def as_dict(self) -> dict:
        return {"rank": self.rank, "world": self.world, "device_ordinal": self.device_ordinal, "nccl_device": self.nccl_device,
                "pci": f"{self.pci_domain:04x}:{self.pci_bus:02x}:{self.pci_device:02x}", "pid": self.pid,
                "host": f"{self.host_hash:016x}", "uuid": bytes(self.uuid).hex(), "arch": self.gcn_arch.decode(errors="replace")}


assert C.sizeof(Peer) == 72


def world_proof(peers, expect_gpus: int) -> dict:
    """What a multi-GPU record may claim, from the identity records the ranks exchanged THROUGH the communicator (`Communicator.peers`;
    dicts as `Peer.as_dict` makes them, in rank order).  n_gpus = the number of ranks RCCL reports, and the record is only valid when
    every rank reports that same count, the ranks are 0..n-1 in order, every rank drives a DIFFERENT physical GPU (host + PCI address,
    uuid where the driver gives one) and the count is what `--gpus` asked for.  Returns {"ok", "rccl_ranks", "distinct_devices",
    "devices", "errors"}; pure function — the same on every rank, so all ranks take the same exit."""
    errors = []
    n = len(peers)
    counts = sorted({int(q["world"]) for q in peers})
    if counts != [n]:
        errors.append(f"ncclCommCount differs from the number of gathered records: counts {counts}, records {n}")
    if [int(q["rank"]) for q in peers] != list(range(n)):
        errors.append(f"ranks out of order: {[int(q['rank']) for q in peers]}")
    phys = [(q["host"], q["pci"]) for q in peers]
    distinct = len(set(phys))
    if distinct != n:
        dup = sorted({x for x in phys if phys.count(x) > 1})
        errors.append(f"{n} ranks on {distinct} distinct devices: {dup} shared")
    uu = [q["uuid"] for q in peers if q.get("uuid") and set(q["uuid"]) != {"0"}]
    if len(uu) == n and len(set(uu)) != n:
        errors.append("two ranks report the same device uuid")
    if n != expect_gpus:
        errors.append(f"--gpus {expect_gpus} but {n} ranks joined the communicator")
    return {"ok": not errors, "rccl_ranks": n, "distinct_devices": distinct,
            "devices": [f'{q["host"][:8]}/{q["pci"]}' for q in peers], "errors": errors}


def prefer_loopback_bootstrap(world: int, env=None) -> bool:
    """RCCL's rendezvous (the "bootstrap") runs over a TCP socket on an interface it picks itself — the first non-loopback one unless
    NCCL_SOCKET_IFNAME says otherwise.  When every rank is known to live on THIS host — a one-rank communicator, or a launcher whose
    MASTER_ADDR is the loopback address (torch.distributed.run --master-addr 127.0.0.1, the contract of bench.py) — the loopback interface
    is the one that cannot be firewalled, unrouted or renamed under a container, so it is named explicitly unless the caller already chose.
    (Data never travels over it: ranks of one node talk over xGMI / shared memory.)  Returns True when it set the variable."""
    import os

    e = os.environ if env is None else env
    if e.get("NCCL_SOCKET_IFNAME"):
        return False
    if world == 1 or e.get("MASTER_ADDR", "") in ("127.0.0.1", "localhost", "::1"):
        e["NCCL_SOCKET_IFNAME"] = "lo"
        return True
    return False


@contextlib.contextmanager
def loopback_bootstrap(world: int, env=None):
    """`prefer_loopback_bootstrap` for the duration of a communicator's rendezvous ONLY: the variable is put back as it was when the block
    ends, so that another RCCL / NCCL user initialised later in this process (a multi-node torch.distributed group, say — torch carries
    its own copy of the library, which reads the environment at ITS first use) does not inherit a loopback bootstrap it never asked for
    (ADVICE r5).  It does not undo what the librccl this library links has already read: within one copy of RCCL the bootstrap interface
    is chosen once per process.  A host that owns its process may call `prefer_loopback_bootstrap` itself instead."""
    import os

    e = os.environ if env is None else env
    did = prefer_loopback_bootstrap(world, e)
    try:
        yield did
    finally:
        if did:
            e.pop("NCCL_SOCKET_IFNAME", None)


class Communicator:
    """RCCL communicator of the C ABI (include/arrow_gpu.h "multi-GPU"): one rank per GPU, used ONLY for the final
    reduce of whole-column statistics.  Nothing like it exists in the reference (single device + queue,
    crates/array/src/gpu_utils/gpu_device.rs:29-33).

    `reduce` = shard-local kernel + all-gather of one 16-byte record per rank + rank-ordered combine on every rank, so
    the result is identical on all ranks and, for f32 Sum over shards of 256^k rows, bit-identical to the reference's
    whole-column tree.  Rendezvous: rank 0 makes the 128-byte id (`Communicator.unique_id()`), the launcher ships it —
    `from_torch` uses an initialised torch.distributed group (gloo or nccl) for that one broadcast, `from_file` a path
    on a shared filesystem; a C++ host passes it between its per-GPU threads directly."""

    def __init__(self, device, rank: int, world: int, unique_id: bytes, timeout_s: float | None = None):
        """Collective: returns once all `world` ranks have arrived.  timeout_s: None = the library default
        (AGPU_COMM_TIMEOUT_MS, 120 s), 0 = wait for ever; on a timeout ArrowErrorGPU is raised and the process should
        exit (the pending RCCL rendezvous cannot be cancelled)."""
        if len(unique_id) != capi.COMM_ID_BYTES:
            raise ValueError("unique id must be 128 bytes")
        self.device, self.rank, self.world = device, rank, world
        h = C.c_void_p()
        idbuf = C.create_string_buffer(unique_id, capi.COMM_ID_BYTES)
        with loopback_bootstrap(world):
            if timeout_s is None:
                capi.call("agpu_comm_init_rank", device._handle, idbuf, rank, world, C.byref(h))
            else:
                capi.call("agpu_comm_init_rank_timeout", device._handle, idbuf, rank, world, int(timeout_s * 1000), C.byref(h))
        self._h = h

    @staticmethod
    def runtime_info() -> str:
        """Which librccl / libamdhip64 this process runs on (path + version)."""
        buf = C.create_string_buffer(1024)
        capi.call("agpu_comm_runtime_info", buf, len(buf))
        return buf.value.decode()

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(capi.COMM_ID_BYTES)
        capi.call("agpu_comm_get_unique_id", buf)
        return buf.raw

    @classmethod
    def single(cls, device) -> "Communicator":
        """World of one rank (no launcher needed): the same RCCL code path a multi-GPU run takes."""
        with loopback_bootstrap(1):  # around the id too: rank 0's listening socket is opened when the id is made
            return cls(device, 0, 1, cls.unique_id())

    @classmethod
    def from_torch(cls, device, group=None) -> "Communicator":
        import torch.distributed as dist

        rank, world = dist.get_rank(group), dist.get_world_size(group)
        with loopback_bootstrap(world):
            box = [cls.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            return cls(device, rank, world, box[0])

    @classmethod
    def from_file(cls, device, rank: int, world: int, path: str, timeout_s: float = 60.0) -> "Communicator":
        """Torch-free rendezvous over a path every rank can see (one node: /tmp or /dev/shm).  See `file_rendezvous`."""
        with loopback_bootstrap(world):
            uid = file_rendezvous(path, rank, world, cls.unique_id if rank == 0 else None, timeout_s)
            comm = cls(device, rank, world, uid, timeout_s=timeout_s)
        file_rendezvous_cleanup(path, rank)
        return comm

    @classmethod
    def from_env(cls, device, timeout_s: float = 120.0) -> "Communicator":
        """One rank per process as a launcher (torch.distributed.run, mpirun, srun, a shell loop) sets it up: RANK /
        WORLD_SIZE (or OMPI_COMM_WORLD_RANK / _SIZE, SLURM_PROCID / SLURM_NTASKS) from the environment, the id through
        `file_rendezvous` at `rendezvous_path_from_env()`.  Nothing of torch is imported."""
        rank, world, _ = ranks_from_env()
        if world == 1:
            return cls.single(device)
        return cls.from_file(device, rank, world, rendezvous_path_from_env(), timeout_s)

    # -- collectives (asynchronous on the pipeline's stream; results are 1-element device buffers)
    def reduce(self, pipeline, op: int, dtype: int, values, validity, n_local: int, out) -> None:
        capi.call("agpu_comm_reduce", self._h, pipeline._handle, op, dtype, _ptr(values), _ptr(validity), n_local, _ptr(out))

    def reduce_sum_f64(self, pipeline, values, validity, n_local: int, out) -> None:
        capi.call("agpu_comm_reduce_sum_f64", self._h, pipeline._handle, _ptr(values), _ptr(validity), n_local, _ptr(out))

    def reduce_stats_f32(self, pipeline, values, validity, n_local: int, out) -> None:
        """sum / min / max / f64 sum of a sharded f32 column with ONE pass over this rank's shard (agpu_comm_reduce_stats_f32): `out` = a
        24-byte agpu_f32_stats record on the device, every field what `reduce` / `reduce_sum_f64` give for that statistic."""
        capi.call("agpu_comm_reduce_stats_f32", self._h, pipeline._handle, _ptr(values), _ptr(validity), n_local, _ptr(out))

    def final_reduce(self, pipeline, op: int, dtype: int, partial, n_local: int, out, f64: bool = False) -> None:
        capi.call("agpu_comm_final_reduce", self._h, pipeline._handle, op, dtype, 1 if f64 else 0, _ptr(partial), n_local, _ptr(out))

    def all_reduce(self, pipeline, op: int, ctype: int, buf, count: int) -> None:
        capi.call("agpu_comm_all_reduce", self._h, pipeline._handle, op, ctype, _ptr(buf), count)

    def barrier(self, pipeline) -> None:
        capi.call("agpu_comm_barrier", self._h, pipeline._handle)

    def sync(self, pipeline) -> None:
        """Host wait for the pipeline's stream WITH the collective deadline (AGPU_COMM_TIMEOUT_MS): the wait to use behind
        `reduce` / `all_reduce` / `final_reduce` — a plain `pipeline.sync()` or a download would block for ever behind a
        collective a dead peer never joins.  On a timeout the device is poisoned (see include/arrow_gpu.h): exit the process."""
        capi.call("agpu_comm_sync", self._h, pipeline._handle)

    @property
    def is_local(self) -> bool:
        """True for a ONE-rank communicator whose RCCL bootstrap did not come up in time: no RCCL behind it, collectives are device
        copies (include/arrow_gpu.h agpu_comm_is_local).  A record that says "RCCL ran" must check this."""
        v = C.c_int32()
        capi.call("agpu_comm_is_local", self._h, C.byref(v))
        return bool(v.value)

    def size(self) -> tuple:
        """(ncclCommCount, ncclCommUserRank, ncclCommCuDevice): what RCCL reports, not what the launcher said."""
        n, r, d = C.c_int32(), C.c_int32(), C.c_int32()
        capi.call("agpu_comm_size", self._h, C.byref(n), C.byref(r), C.byref(d))
        return n.value, r.value, d.value

    def peers(self, pipeline) -> list:
        """Collective: every rank's identity record (`Peer.as_dict`), gathered through the communicator, in rank order."""
        arr = (Peer * self.world)()
        distinct = C.c_int32()
        capi.call("agpu_comm_peers", self._h, pipeline._handle, arr, self.world, C.byref(distinct))
        return [q.as_dict() for q in arr]

    def close(self) -> None:
        if getattr(self, "_h", None):
            capi.lib().agpu_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, int):
        return C.c_void_p(x)
    return C.c_void_p(x.ptr)
