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
