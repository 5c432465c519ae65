"""CPU: the N>1 path — shard planning + final reduce over torch.distributed — with world_size 2 on the gloo backend.
Per-shard partials come from the ORACLE here (no GPU in this container); on the GPU box bench.py feeds the same
`sharding.final_reduce` with partials computed by the HIP kernels, backend "nccl" (= RCCL over xGMI)."""
import os
import socket

import numpy as np
import pytest

from arrow_gpu_amd import sharding


def test_shard_plan_properties():
    for total in (0, 1, 511, 512, 513, 10_000, 1_000_000_007, 8_000_000_000):
        for world in (1, 2, 3, 4, 8):
            shards = sharding.all_shards(total, world)
            assert shards[0].row0 == 0 and shards[-1].row_end == total
            for a, b in zip(shards, shards[1:]):
                assert a.row_end == b.row0          # contiguous, no gap, no overlap
            for s in shards[:-1]:
                assert s.row_end % 512 == 0 or s.row_end == total  # whole bitmap words / 16-byte vectors per rank
            sizes = [s.rows for s in shards]
            assert max(sizes) - min(sizes) <= 1024  # ≤ one 512-row chunk of imbalance + the ragged last chunk
    s = sharding.shard_rows(8_000_000_000, 8, 3)
    assert (s.row0, s.rows) == (3_000_000_000, 1_000_000_000)  # config 5: 8 × 1e9-row shards
    with pytest.raises(ValueError):
        sharding.shard_rows(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    import torch
    import torch.distributed as dist

    import oracle as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = sharding.shard_rows(total, world, rank)
        x = O.synth_f32(sh.rows, 20250418, sh.row0, -1.0, 1.0)        # this rank's shard of the column
        bits = O.synth_bits(sh.rows, 7, sh.row0, 0.9)
        s = torch.tensor([O.reduce_sum_f64(x)], dtype=torch.float64)
        mn = torch.tensor([float(O.reduce(O.RED_MIN, O.F32, x))], dtype=torch.float32)
        mx = torch.tensor([float(O.reduce(O.RED_MAX, O.F32, x))], dtype=torch.float32)
        cnt = torch.tensor([O.bitmap_popcount(bits, sh.rows)], dtype=torch.int64)
        sharding.final_reduce(s, mn, mx, cnt)
        q.put((rank, s.item(), mn.item(), mx.item(), cnt.item(), sh.row0, sh.rows))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 1_000_003), (2, 512 * 4096)])
def test_final_reduce_world2_gloo(world, total):
    import torch.multiprocessing as mp

    import oracle as O

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    whole = O.synth_f32(total, 20250418, 0, -1.0, 1.0)
    bits = O.synth_bits(total, 7, 0, 0.9)
    exp_sum = float(np.sum(whole.astype(np.float64)))
    for rank, s, mn, mx, cnt, row0, rows in res:
        assert abs(s - exp_sum) <= 1e-9 * float(np.sum(np.abs(whole.astype(np.float64))))
        assert mn == float(whole.min()) and mx == float(whole.max())          # exact
        assert cnt == O.bitmap_popcount(bits, total)                            # exact
    assert sorted(r[5] for r in res) == [s.row0 for s in sharding.all_shards(total, world)]


def test_final_reduce_single_process_is_identity():
    import torch

    s = torch.tensor([1.5], dtype=torch.float64)
    out = sharding.final_reduce(s, None, None, None)
    assert out[0] is s and s.item() == 1.5


# ---- the C ABI's protocol (agpu_comm_reduce: all-gather of one {statistic, n_local} record per rank, rank-ordered combine)
# played by two gloo processes with the oracle as the shard-local kernel: every rank must arrive at the spec's value.
def _record_worker(rank, world, port, shard_rows_list, case, q):
    import torch.distributed as dist

    import oracle as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        row0 = sum(shard_rows_list[:rank])
        x = O.synth_f32(shard_rows_list[rank], 20250418, row0, -1000.0, 1000.0)
        if case == "nan" and len(x):
            x[:] = np.nan if rank == 0 else x
        out = {}
        for op in (O.RED_SUM, O.RED_MIN, O.RED_MAX):
            mine = (float(O.reduce(op, O.F32, x)), len(x))
            gathered = [None] * world
            dist.all_gather_object(gathered, mine)
            out[op] = float(O.combine_records(op, O.F32, gathered))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("rows,case", [([65536, 65536], "plain"), ([1000, 0], "empty"), ([4096, 4096], "nan"), ([70_001, 33], "plain")])
def test_record_gather_protocol_world2_matches_the_sharded_spec(rows, case):
    import torch.multiprocessing as mp

    import oracle as O

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_record_worker, args=(r, 2, port, rows, case, q)) for r in range(2)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=120) for _ in range(2))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    shards, row0 = [], 0
    for r, n in enumerate(rows):
        x = O.synth_f32(n, 20250418, row0, -1000.0, 1000.0)
        if case == "nan" and r == 0:
            x[:] = np.nan
        shards.append(x)
        row0 += n
    for op in (O.RED_SUM, O.RED_MIN, O.RED_MAX):
        exp = float(O.sharded_reduce(op, O.F32, shards))
        for rank in (0, 1):
            got = res[rank][op]
            assert (np.isnan(got) and np.isnan(exp)) or np.float32(got).view(np.uint32) == np.float32(exp).view(np.uint32), (op, rank, got, exp)
    if rows == [65536, 65536]:  # shards of 256^2 rows: the sharded Sum IS the reference's whole-column tree
        whole = np.concatenate(shards)
        assert np.float32(res[0][O.RED_SUM]).view(np.uint32) == np.float32(O.reduce(O.RED_SUM, O.F32, whole)).view(np.uint32)


def test_record_batches_are_sharded_as_contiguous_runs_in_rank_order():
    """sharding.shard_batches: every batch goes to exactly one rank, runs are contiguous and ordered, row counts are balanced
    to within one batch — for equal, ragged and degenerate batch lists"""
    from arrow_gpu_amd.sharding import shard_batches

    import numpy as np

    rng = np.random.default_rng(2)
    for rows in ([1000] * 16, [5, 5, 5], [7], [], list(rng.integers(1, 100_000, 37)), [0, 0, 10, 0, 10, 10, 0]):
        rows = [int(x) for x in rows]
        for world in (1, 2, 3, 8):
            runs = [shard_batches(rows, world, r) for r in range(world)]
            assert [b for run in runs for b in run] == list(range(len(rows)))  # a partition, in order
            per_rank = [sum(rows[b] for b in run) for run in runs]
            if rows and sum(rows):
                assert max(per_rank) - sum(rows) / world <= max(rows)


# ---- the record that makes an N > 1 bench line self-proving: `sharding.world_proof` over the identity records
# the ranks exchange through the communicator (`agpu_comm_peers` on the GPU; gloo's all_gather_object as the stand-in here)
def _peer(rank, world, pci_bus, host="00c0ffee00c0ffee", uuid=None):
    return {"rank": rank, "world": world, "device_ordinal": rank, "nccl_device": rank, "pci": f"0000:{pci_bus:02x}:00", "pid": 1000 + rank,
            "host": host, "uuid": uuid if uuid is not None else f"{pci_bus:032x}", "arch": "gfx950"}


def test_world_proof_accepts_n_distinct_devices_and_nothing_else():
    from arrow_gpu_amd.sharding import world_proof

    good = [_peer(r, 8, 0x10 + r) for r in range(8)]
    pr = world_proof(good, 8)
    assert pr["ok"] and pr["rccl_ranks"] == 8 and pr["distinct_devices"] == 8 and len(pr["devices"]) == 8 and not pr["errors"]
    assert world_proof([_peer(0, 1, 5)], 1)["ok"]
    # the launcher was asked for 8 but 4 ranks joined
    pr = world_proof([_peer(r, 4, 0x10 + r) for r in range(4)], 8)
    assert not pr["ok"] and pr["rccl_ranks"] == 4 and any("--gpus 8" in e for e in pr["errors"])
    # two ranks on one physical GPU (same host + PCI address): 8 ranks, 7 devices
    shared = [_peer(r, 8, 0x10 + (r if r != 7 else 0)) for r in range(8)]
    pr = world_proof(shared, 8)
    assert not pr["ok"] and pr["distinct_devices"] == 7 and any("distinct devices" in e for e in pr["errors"])
    # the same PCI address on ANOTHER host is another device
    two_nodes = [_peer(r, 2, 0x10, host=f"{r:016x}", uuid=f"{r + 1:032x}") for r in range(2)]
    assert world_proof(two_nodes, 2)["ok"]
    # a rank whose RCCL communicator has another size than the number of records (two communicators mixed up)
    odd = [_peer(0, 2, 0x10), _peer(1, 3, 0x11)]
    assert not world_proof(odd, 2)["ok"]
    # ranks out of order / duplicated
    assert not world_proof([_peer(1, 2, 0x10), _peer(0, 2, 0x11)], 2)["ok"]
    assert not world_proof([_peer(0, 2, 0x10), _peer(0, 2, 0x11)], 2)["ok"]
    # same uuid behind two PCI addresses (a partitioned GPU seen twice)
    assert not world_proof([_peer(0, 2, 0x10, uuid="ab" * 16), _peer(1, 2, 0x11, uuid="ab" * 16)], 2)["ok"]
    # drivers that report an all-zero uuid are not held against the run
    assert world_proof([_peer(0, 2, 0x10, uuid="0" * 32), _peer(1, 2, 0x11, uuid="0" * 32)], 2)["ok"]


def _proof_worker(rank, world, port, same_device, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = _peer(dist.get_rank(), dist.get_world_size(), 0x20 if same_device else 0x20 + rank)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        q.put((rank, sharding.world_proof(gathered, world)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("same_device", [False, True])
def test_world_proof_world2_gloo_every_rank_reaches_the_same_verdict(same_device):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_proof_worker, args=(r, 2, port, same_device, q)) for r in range(2)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=120) for _ in range(2))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert res[0] == res[1]                       # a pure function of the gathered records: all ranks take the same exit
    assert res[0]["ok"] == (not same_device) and res[0]["rccl_ranks"] == 2
    assert res[0]["distinct_devices"] == (1 if same_device else 2)
