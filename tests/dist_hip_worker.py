"""Child process of test_two_processes_sharded_msm_on_one_gpu (not collected by pytest): one rank of a 2-rank gloo
group, its own HipEngine on GPU 0 holding ITS SRS segment, running the product's sharded_msm -- through host bytes,
through a device-resident slot, and through the STREAM-CHAINED collective step (DeviceGather: the lane's partial goes
into a device tensor, the all_gather runs on device tensors behind torch's current stream, the lane sums the gathered
partials; gloo moves device tensors too, which lets two ranks share this one GPU where RCCL cannot).  Also the failure
path: a collective that raises must not strand the lane (kzg_msm_cancel).  Prints the hex result.

    RANK=r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/dist_hip_worker.py LOG2_N_PER_RANK TAU_HEX
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402

from zkp_subnet_amd import HipEngine  # noqa: E402
from zkp_subnet_amd.distributed import sharded_msm  # noqa: E402

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
lg, tx = int(sys.argv[1]), int(sys.argv[2], 16)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
n = 1 << lg
dist.init_process_group("gloo", rank=rank, world_size=world)
eng = HipEngine(0)
eng.gen_srs(tx, 1, lg, 0, factors=[pow(tx, rank * n, R)])        # segment [rank*n, (rank+1)*n) of [tau^j]G
raw = np.random.default_rng(7000 + rank).integers(0, 256, size=(n, 32), dtype=np.uint8)
raw[:, 0] &= 0x3F
shard = raw.tobytes()
a = sharded_msm(eng, shard)                                        # host scalars -> partial -> all_gather -> sum
eng.upload_fr(0, shard, False)
b = sharded_msm(eng, slot=0, n=n)                                  # device-resident scalars
assert a == b
import torch  # noqa: E402

from zkp_subnet_amd.distributed import DeviceGather  # noqa: E402

torch.cuda.set_device(0)
g = DeviceGather(eng)
for _ in range(3):
    assert g.msm(0, n, 0) == a                                     # chained: lane -> current stream -> collective -> lane
assert g.msm_blocking(0, n, 0) == a
real = dist.all_gather_into_tensor


def broken(*args, **kw):
    raise RuntimeError("simulated collective failure")


dist.all_gather_into_tensor = broken
try:
    g.msm(0, n, 0)
    raise SystemExit("the failing collective did not propagate")
except RuntimeError:
    pass
dist.all_gather_into_tensor = real
eng.upload_fr(1, shard, False)                                     # exclusive call: fails with E_BUSY if the lane stayed parked
assert g.msm(1, n, 0) == a
eng.close()
dist.barrier()
dist.destroy_process_group()
print(a.hex())
