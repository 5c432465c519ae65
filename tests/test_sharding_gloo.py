"""CPU: the N > 1 path with world_size 2 on the gloo backend — shard planning, the C ABI's record-gather protocol of the final reduce
(agpu_comm_reduce) with the ORACLE as the shard-local kernel, and what makes a bench line of any world self-proving: the world proof
and bench.py's per-rank parity / final-reduce checks (the numbers travel by a SUM all-reduce of one slot per rank, exactly as bench.py
sends them through agpu_comm_all_reduce on the GPU box; gloo's all_reduce stands in for RCCL here)."""
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


# ---- bench.py at world > 1: every rank checks windows of ITS shard against the oracle, the verdicts are AND-ed through
# the communicator; the collectives' sum / min / max / f64 sum are checked on every rank against the oracle's rank-ordered combine of the
# per-rank local statistics, gathered by a SUM all-reduce of one slot per rank (bench.slot_vector / reduce_records_from_gathered /
# verify_final_reduce).  Played here by two gloo processes: the shard-local "kernels" are the oracle's, a corrupted rank must be caught.
def _bench_protocol_worker(rank, world, port, shard_rows_list, corrupt, q):
    import sys

    import torch
    import torch.distributed as dist

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import oracle as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def across_ranks(values, op):
            t = torch.tensor(values, dtype=torch.float64)
            dist.all_reduce(t, op=op)
            return [float(x) for x in t]

        n, row0 = shard_rows_list[rank], sum(shard_rows_list[:rank])
        # (1) the headline windows of THIS rank's shard, at its own row0
        cnt = min(n, 4096)
        add = O.binary(O.OP_ADD, O.F32, O.synth_f32(cnt, bench.SEED, row0, -1000.0, 1000.0), O.synth_f32(cnt, bench.SEED + 1, row0, -1000.0, 1000.0))
        eq = O.compare(O.CMP_EQ, O.I32, O.synth_i32(cnt, bench.SEED + 2, row0, 1024), O.synth_i32(cnt, bench.SEED + 3, row0, 1024))
        vd = O.bitmap_binary(O.OP_AND, O.synth_bits(cnt, bench.SEED + 4, row0, 0.9), O.synth_bits(cnt, bench.SEED + 5, row0, 0.9), cnt)
        if corrupt == "window" and rank == 1:
            add = add.copy()
            add[7] = np.float32(1.0)
        ok_local = bench.check_headline_windows([{"row": row0, "rows": cnt, "add": add, "eq_bits": eq, "eq_validity": vd}])
        bad_ranks = int(round(across_ranks([0.0 if ok_local else 1.0], dist.ReduceOp.SUM)[0]))
        # (2) the final reduce: local statistics → gathered records → the "collective's" result on every rank → check
        x = O.synth_f32(n, bench.SEED, row0, -1000.0, 1000.0)
        loc = {"sum": O.reduce(O.RED_SUM, O.F32, x), "min": O.reduce(O.RED_MIN, O.F32, x), "max": O.reduce(O.RED_MAX, O.F32, x),
               "sum_f64": O.reduce_sum_f64(x)}
        b64 = int(np.array([loc["sum_f64"]], np.float64).view(np.uint64)[0])
        mine = [int(np.array([loc[k]], np.float32).view(np.uint32)[0]) for k in ("sum", "min", "max")] + [b64 & 0xFFFFFFFF, b64 >> 32, n]
        gathered = across_ranks(bench.slot_vector(rank, world, mine), dist.ReduceOp.SUM)
        recs = bench.reduce_records_from_gathered(gathered, world)
        got = {"sum": O.combine_records(O.RED_SUM, O.F32, recs["sum"]), "min": O.combine_records(O.RED_MIN, O.F32, recs["min"]),
               "max": O.combine_records(O.RED_MAX, O.F32, recs["max"]), "sum_f64": sum(float(v) for v, k in recs["sum_f64"] if k)}
        if corrupt == "reduce" and rank == 0:
            got["max"] = np.float32(got["max"]) + np.float32(1.0)
        detail = bench.verify_final_reduce(recs, got)
        unverified = int(round(across_ranks([0.0 if all(detail.values()) else 1.0], dist.ReduceOp.SUM)[0]))
        q.put((rank, bad_ranks, unverified, [(float(v), int(k)) for v, k in recs["sum"]], float(got["sum"])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("rows,corrupt", [([65536, 65536], None), ([70_001, 33], None), ([5000, 0], None), ([65536, 65536], "window"),
                                          ([65536, 65536], "reduce")])
def test_bench_per_rank_parity_and_final_reduce_check_world2_gloo(rows, corrupt):
    import torch.multiprocessing as mp

    import oracle as O

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_protocol_worker, args=(r, 2, port, rows, corrupt, q)) for r in range(2)]
    [p.start() for p in procs]
    res = {r[0]: r[1:] for r in (q.get(timeout=120) for _ in range(2))}
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert res[0][0] == res[1][0] == (1 if corrupt == "window" else 0)   # every rank knows how many ranks failed their windows
    assert res[0][1] == res[1][1] == (1 if corrupt == "reduce" else 0)   # … and how many hold a result that is not the oracle's combine
    assert res[0][2] == res[1][2]                                          # the gathered records are the same table on every rank
    shards, row0 = [], 0
    for n in rows:
        shards.append(O.synth_f32(n, 20250418, row0, -1000.0, 1000.0))
        row0 += n
    assert [k for _, k in res[0][2]] == rows
    for (v, _), sh in zip(res[0][2], shards):  # slot r of the gathered table IS rank r's local sum, bit for bit
        assert np.float32(v).view(np.uint32) == np.float32(O.reduce(O.RED_SUM, O.F32, sh)).view(np.uint32)
    assert np.float32(res[0][3]).view(np.uint32) == np.float32(O.sharded_reduce(O.RED_SUM, O.F32, shards)).view(np.uint32)


def test_slot_vector_round_trip_is_exact_for_bit_patterns():
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    world = 8
    rng = np.random.default_rng(5)
    per_rank = [[int(x) for x in rng.integers(0, 2 ** 32, 5)] + [int(rng.integers(0, 2 ** 40))] for _ in range(world)]
    total = np.zeros(6 * world)
    for r in range(world):
        total += np.array(bench.slot_vector(r, world, per_rank[r]))
    recs = bench.reduce_records_from_gathered(list(total), world)
    for r in range(world):
        s, mn, mx, lo, hi, n = per_rank[r]
        assert np.float32(recs["sum"][r][0]).view(np.uint32) == s or np.isnan(recs["sum"][r][0])
        assert recs["min"][r][1] == recs["sum_f64"][r][1] == n
        assert int(np.array([recs["sum_f64"][r][0]], np.float64).view(np.uint64)[0]) == ((hi << 32) | lo)
