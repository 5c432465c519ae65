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

from dataclasses import dataclass

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
