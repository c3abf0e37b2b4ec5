"""world_size-2 gloo test of the N>1 path (runs on CPU): SRS-segment sharding, the all_gather of 192-byte partials
and the final sum, with the oracle engine computing the per-rank partials.  On the GPU box the same
zkp_subnet_amd.distributed code runs over RCCL with HipEngine partials (bench.py --gpus N)."""
import os
import random
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import bls12_381 as o
from oracle import cpu as oc
from tests.oracle_engine import OracleEngine
from zkp_subnet_amd.distributed import all_gather_partials, shard_range, sharded_msm


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 1 << 20, (1 << 20) + 3):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_world_of_one_needs_no_process_group():
    p = bytes(range(192))
    assert all_gather_partials(p) == [p]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, srs, scalars, expect, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        lo, hi = shard_range(n, rank, world)
        eng = OracleEngine()
        eng.load_srs(srs[96 * lo:96 * hi], 0, 0)       # this rank's SRS segment only
        got = sharded_msm(eng, scalars[32 * lo:32 * hi], 0)
        dist.barrier()
        q.put((rank, got == expect))
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_msm_gloo(world):
    rnd = random.Random(world)
    n = 37
    tx = rnd.randrange(1, o.R)
    srs = oc.srs_gen(tx.to_bytes(32, "big"), (1).to_bytes(32, "big"), 6, 0, 0)[: 96 * n]
    sc = [rnd.randrange(o.R) for _ in range(n)]
    scalars = o.fr_to_be32(sc)
    expect = oc.msm(srs, scalars)
    assert expect == oc.g1_mul_gen(o.poly_eval(sc, tx).to_bytes(32, "big"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, srs, scalars, expect, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(r, True) for r in range(world)], results


def _ctl_worker(rank, world, port, q):
    import datetime

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch

    import bench

    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        ctl = bench.Ctl(torch, dist, rank, world, True, 20)
        out = {"store": ctl.store is not None}
        ctl.barrier()
        out["max"] = ctl.max_over_ranks(float(rank + 1))
        out["gather"] = [b.hex() for b in ctl.gather_bytes(bytes([rank]) * 4)]
        out["all_ok"] = ctl.agree("p1", True)
        out["one_bad"] = ctl.agree("p2", rank != 1, "boom on 1" if rank == 1 else "")
        # a rank that never reports counts as failed after the timeout -- nobody waits for ever
        ctl2 = bench.Ctl(torch, dist, rank, world, True, 2)
        if rank == 0:
            out["silent_peer"] = ctl2.agree("p3", True)
        # the fault injector of the bench's tests
        os.environ["BENCH_FAULT"] = "msm26_setup:1,comm_init:0"
        hits = []
        for phase in ("msm26_setup", "comm_init", "msm26_step"):
            try:
                bench.inject(phase, rank)
            except bench.Fault:
                hits.append(phase)
        out["faults"] = hits
        q.put((rank, out))
        ctl.barrier()
    finally:
        dist.destroy_process_group()


def test_bench_control_plane_status_travels_through_the_store_world2():
    """bench.py's control plane (VERDICT r4 task 1b): collectives on a gloo group, phase STATUS through the rendezvous store --
    a failure on one rank is seen by every rank without anybody entering a collective, and a rank that never reports is a
    failure after the timeout, not a hang."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 300) + 350
    procs = [ctx.Process(target=_ctl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in range(world):
        o_ = got[r]
        assert o_["store"] and o_["max"] == 2.0 and o_["gather"] == ["00000000", "01010101"]
        assert o_["all_ok"] == (True, {}) and o_["one_bad"] == (False, {1: "boom on 1"})
    ok, bad = got[0]["silent_peer"]
    assert not ok and list(bad) == [1] and "no status within 2 s" in bad[1]
    assert got[0]["faults"] == ["comm_init"] and got[1]["faults"] == ["msm26_setup"]
