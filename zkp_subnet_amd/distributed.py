"""Multi-GPU MSM: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

The MSM shards by SRS segment: rank g owns points / scalars [g*n/G, (g+1)*n/G), reduces them to ONE
192-byte partial sum on its GPU, and the only exchange on the data path is an all_gather of those partials
(G x 192 B: latency-bound, a ring all-reduce is neither needed nor expressible -- RCCL has no EC-add reduce op),
followed by G-1 point additions + affine + compression on every rank.  Pianist segments (one worker row per GPU)
need no exchange at all."""
from __future__ import annotations

from typing import List, Optional, Tuple

PARTIAL_BYTES = 192


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition [lo, hi) of n items; sizes differ by at most one."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_partials(partial: bytes, group=None) -> List[bytes]:
    """all_gather of one 192-byte partial per rank.  Uses the default process group's backend: device tensors for
    nccl/RCCL, host tensors for gloo.  Without an initialised process group: world of one."""
    import torch
    import torch.distributed as dist

    assert len(partial) == PARTIAL_BYTES
    if not (dist.is_available() and dist.is_initialized()):
        return [partial]
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")   # host bytes in, host bytes out
    src = torch.frombuffer(bytearray(partial), dtype=torch.uint8).to(dev)
    out = torch.empty(world * PARTIAL_BYTES, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, src, group=group)
    raw = out.cpu().numpy().tobytes()
    return [raw[i * PARTIAL_BYTES:(i + 1) * PARTIAL_BYTES] for i in range(world)]


class LibraryGather:
    """The multi-GPU step with the collective INSIDE the library (`kzg_comm_init` / `kzg_msm_sharded`): the 192-byte
    partial, RCCL's all_gather and the sum run on the engine's own lane stream -- no torch tensor, no foreign stream, one
    host wait.  torch.distributed (any backend) is only the rendezvous: rank 0 draws the RCCL unique id and the process
    group's object broadcast carries its 128 bytes.  `DeviceGather` below is the torch-collective form kept as the A/B.

    `init_timeout_s`: ncclCommInitRank blocks in native code until every rank has joined; the LIBRARY bounds it
    (kzg_comm_init_bounded: the rendezvous and a first checked all_gather run on a helper thread that holds nothing of the
    context), so a rank whose peers never arrive raises TimeoutError here and its engine keeps working -- every other call,
    `close()` and interpreter exit included."""

    def __init__(self, engine, group=None, timeout_ms: int = 0, init_timeout_s: float = 120.0):
        import torch.distributed as dist

        from ._native import KZG_E_COMM, KzgError

        self.engine = engine
        self.world = dist.get_world_size(group)
        rank = dist.get_rank(group)
        box = [None]
        if rank == 0:
            try:
                box[0] = engine.comm_unique_id()
            except Exception as e:               # noqa: BLE001 -- every rank must learn of it, or they wait for the id
                box[0] = e
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if isinstance(box[0], Exception):
            raise box[0]
        try:
            engine.comm_init(box[0], rank, self.world, timeout_ms, init_timeout_ms=max(1, int(init_timeout_s * 1000)))
        except KzgError as e:
            if e.code == KZG_E_COMM and "did not all join" in str(e):
                raise TimeoutError(f"kzg_comm_init: the {self.world} ranks did not all join within {init_timeout_s:.0f} s") from e
            raise
        self.info = engine.comm_info()

    def msm(self, slot: int, n: int, srs_offset: int = 0) -> bytes:
        return self.engine.msm_sharded(slot, n, srs_offset)

    def close(self) -> None:
        self.engine.comm_destroy()


class DeviceGather:
    """The collective step with no host round trip for the partials: the engine writes its 192-byte partial into a
    torch device tensor, RCCL all_gathers device-to-device, the engine sums the gathered tensor on the GPU.  Tensors are
    allocated once.  Backend nccl (= RCCL); gloo also moves device tensors (through the host), which is how the tests run
    this step with two real ranks on one GPU."""

    def __init__(self, engine, group=None):
        import torch
        import torch.distributed as dist

        self.engine, self.group, self.dist, self.torch = engine, group, dist, torch
        self.world = dist.get_world_size(group)
        dev = torch.device("cuda", torch.cuda.current_device())
        self.src = torch.zeros(PARTIAL_BYTES, dtype=torch.uint8, device=dev)
        self.out = torch.zeros(self.world * PARTIAL_BYTES, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()

    def msm(self, slot: int, n: int, srs_offset: int = 0) -> bytes:
        """partial -> all_gather -> sum, chained on the device: the engine's lane, torch's current stream (RCCL behind it)
        and the lane again hand over through events; the host blocks once, for the result."""
        stream = self.torch.cuda.current_stream().cuda_stream
        ticket = self.engine.msm_sharded_begin(slot, n, srs_offset, self.src.data_ptr(), stream)
        try:
            self.dist.all_gather_into_tensor(self.out, self.src, group=self.group)
        except BaseException:
            # the collective raised (RCCL timeout, peer death): nobody will call _finish -- free the lane, or it stays
            # parked under the ticket and every exclusive call returns E_BUSY from then on
            self.engine.msm_cancel(ticket)
            raise
        return self.engine.msm_sharded_finish(ticket, self.out.data_ptr(), self.world, stream)

    def msm_blocking(self, slot: int, n: int, srs_offset: int = 0) -> bytes:
        """The same step through the blocking entry points (three host synchronisations)."""
        self.engine.msm_partial_resident_dev(slot, n, srs_offset, self.src.data_ptr())   # complete on return
        self.dist.all_gather_into_tensor(self.out, self.src, group=self.group)
        self.torch.cuda.current_stream().synchronize()                                   # RCCL done before the sum reads
        return self.engine.g1_sum_dev(self.out.data_ptr(), self.world)


def sharded_msm(engine, scalars_shard_be32: Optional[bytes] = None, srs_offset: int = 0, slot: Optional[int] = None,
                n: Optional[int] = None, group=None) -> bytes:
    """This rank's shard -> partial -> all_gather -> sum.  Every rank returns the same 48-byte compressed point.
    Either pass the shard's scalars (host bytes) or a device-resident slot + n."""
    if slot is not None:
        partial = engine.msm_partial_resident(slot, n, srs_offset)
    else:
        partial = engine.msm_partial(scalars_shard_be32, srs_offset)
    return engine.g1_sum(b"".join(all_gather_partials(partial, group)))
