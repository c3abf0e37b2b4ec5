"""Child process of test_gpu_multi.py: a ONE-rank RCCL group on cuda:0.  The collective step of the sharded MSM through
the stream-chained entry points (kzg_msm_sharded_begin / _finish: lane -> torch's stream -> RCCL -> lane, one host
synchronisation) must return the bytes of the blocking pair and of the plain single-GPU MSM.  Prints one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from zkp_subnet_amd import HipEngine  # noqa: E402
from zkp_subnet_amd.distributed import DeviceGather  # noqa: E402

port = sys.argv[1]
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                        device_id=torch.device("cuda", 0))
out = {}
for lg in (6, 12, 16):
    n = 1 << lg
    eng = HipEngine(0)
    eng.gen_srs(0x51AB1E + lg, 1, lg, 0)
    raw = np.random.default_rng(lg).integers(0, 256, size=(n, 32), dtype=np.uint8)
    raw[:, 0] &= 0x3F
    eng.upload_fr(0, raw.tobytes(), False)
    g = DeviceGather(eng)
    plain = eng.msm_resident(0, n, 0)
    chained = [g.msm(0, n, 0) for _ in range(3)]
    blocking = g.msm_blocking(0, n, 0)
    # a segment of the SRS (offset, shorter length), as a rank > 0 would hold
    seg_plain = eng.msm_resident(0, n // 2, n // 4)
    seg_chained = g.msm(0, n // 2, n // 4)
    out[str(lg)] = {"plain": plain.hex(), "chained_equal": all(c == plain for c in chained), "blocking_equal": blocking == plain,
                    "segment_equal": seg_chained == seg_plain}
    eng.close()
dist.barrier()
dist.destroy_process_group()
print(json.dumps(out))
