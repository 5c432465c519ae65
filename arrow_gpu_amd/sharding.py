"""Chunk-sharding of a column across the GPUs of one node + the final reduce of whole-column statistics.

Not in the reference (it is single-device, single-queue: SURVEY §5); this is north_star's multi-GPU rule:
one process per GPU, every rank owns ONE contiguous row range of the column, element-wise / compare / cast / bitmap
kernels need no communication (outputs stay sharded), and only sum / min / max / popcount finish with a collective of
ONE element per statistic over `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the
CPU tests).  The message is 4–8 bytes, so the collective is latency-bound and independent of the xGMI link rate.

Shard boundaries fall on multiples of 512 rows so every rank owns whole 64-bit bitmap words and 2 KiB-aligned f32
spans (16-byte vector path, no split validity words).
"""
from __future__ import annotations

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


def final_reduce(sum_t=None, min_t=None, max_t=None, count_t=None, group=None):
    """In-place final reduce of per-shard partials held in 1-element tensors (any device / backend).

    sum_t: float64 partial sums (agpu_reduce_sum_f64) → SUM;  min_t / max_t → MIN / MAX;  count_t: int64 → SUM.
    With a single process (no process group) this is the identity.  Returns the tensors.
    """
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return sum_t, min_t, max_t, count_t
    if sum_t is not None:
        dist.all_reduce(sum_t, op=dist.ReduceOp.SUM, group=group)
    if min_t is not None:
        dist.all_reduce(min_t, op=dist.ReduceOp.MIN, group=group)
    if max_t is not None:
        dist.all_reduce(max_t, op=dist.ReduceOp.MAX, group=group)
    if count_t is not None:
        dist.all_reduce(count_t, op=dist.ReduceOp.SUM, group=group)
    return sum_t, min_t, max_t, count_t


class Communicator:
    """RCCL communicator of the C ABI (include/arrow_gpu.h "multi-GPU"): one rank per GPU, used ONLY for the final
    reduce of whole-column statistics.  Nothing like it exists in the reference (single device + queue,
    crates/array/src/gpu_utils/gpu_device.rs:29-33).

    `reduce` = shard-local kernel + all-gather of one 16-byte record per rank + rank-ordered combine on every rank, so
    the result is identical on all ranks and, for f32 Sum over shards of 256^k rows, bit-identical to the reference's
    whole-column tree.  Rendezvous: rank 0 makes the 128-byte id (`Communicator.unique_id()`), the launcher ships it —
    `from_torch` uses an initialised torch.distributed group (gloo or nccl) for that one broadcast, `from_file` a path
    on a shared filesystem; a C++ host passes it between its per-GPU threads directly."""

    def __init__(self, device, rank: int, world: int, unique_id: bytes):
        if len(unique_id) != capi.COMM_ID_BYTES:
            raise ValueError("unique id must be 128 bytes")
        self.device, self.rank, self.world = device, rank, world
        h = C.c_void_p()
        idbuf = C.create_string_buffer(unique_id, capi.COMM_ID_BYTES)
        capi.call("agpu_comm_init_rank", device._handle, idbuf, rank, world, C.byref(h))
        self._h = h

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(capi.COMM_ID_BYTES)
        capi.call("agpu_comm_get_unique_id", buf)
        return buf.raw

    @classmethod
    def single(cls, device) -> "Communicator":
        """World of one rank (no launcher needed): the same RCCL code path a multi-GPU run takes."""
        return cls(device, 0, 1, cls.unique_id())

    @classmethod
    def from_torch(cls, device, group=None) -> "Communicator":
        import torch.distributed as dist

        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(device, rank, world, box[0])

    @classmethod
    def from_file(cls, device, rank: int, world: int, path: str, timeout_s: float = 60.0) -> "Communicator":
        import os
        import time

        if rank == 0:
            tmp = path + ".tmp"
            with open(tmp, "wb") as f:
                f.write(cls.unique_id())
            os.replace(tmp, path)
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > timeout_s:
                raise TimeoutError(f"no communicator id at {path}")
            time.sleep(0.01)
        with open(path, "rb") as f:
            return cls(device, rank, world, f.read())

    # -- collectives (asynchronous on the pipeline's stream; results are 1-element device buffers)
    def reduce(self, pipeline, op: int, dtype: int, values, validity, n_local: int, out) -> None:
        capi.call("agpu_comm_reduce", self._h, pipeline._handle, op, dtype, _ptr(values), _ptr(validity), n_local, _ptr(out))

    def reduce_sum_f64(self, pipeline, values, validity, n_local: int, out) -> None:
        capi.call("agpu_comm_reduce_sum_f64", self._h, pipeline._handle, _ptr(values), _ptr(validity), n_local, _ptr(out))

    def final_reduce(self, pipeline, op: int, dtype: int, partial, n_local: int, out, f64: bool = False) -> None:
        capi.call("agpu_comm_final_reduce", self._h, pipeline._handle, op, dtype, 1 if f64 else 0, _ptr(partial), n_local, _ptr(out))

    def all_reduce(self, pipeline, op: int, ctype: int, buf, count: int) -> None:
        capi.call("agpu_comm_all_reduce", self._h, pipeline._handle, op, ctype, _ptr(buf), count)

    def barrier(self, pipeline) -> None:
        capi.call("agpu_comm_barrier", self._h, pipeline._handle)

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
