"""Child process of test_gpu_units.py: what kzg_runtime_info reports in a fresh process under a given GPU_MAX_HW_QUEUES, and
that results do not depend on it.  argv[1] = "torch-first": initialise HIP through torch BEFORE the library is loaded.
Prints one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "torch-first":
    import torch

    torch.zeros(1, device="cuda:0")         # the HIP runtime is live (and has read its environment) before our library loads
import numpy as np  # noqa: E402

from zkp_subnet_amd import HipEngine  # noqa: E402

eng = HipEngine(0)
lg = 12
raw = np.random.default_rng(5).integers(0, 256, size=(1 << lg, 32), dtype=np.uint8)
raw[:, 0] &= 0x3F
eng.gen_srs(0x1234ABCD, 1, lg, 0)
eng.upload_fr(0, raw.tobytes(), False)
t = [eng.msm_submit(0, 1 << lg, 0) for _ in range(2)]          # two requests in flight: two lanes whatever the queues
res = [eng.msm_wait(x).hex() for x in t]
out = {"info": eng.runtime_info(), "msm": eng.msm(raw.tobytes(), 0).hex(), "tickets": res}
eng.close()
# Client.start says so when the lanes do not overlap (and only then)
import logging  # noqa: E402

from zkp_subnet_amd import Client  # noqa: E402

seen = []


class Grab(logging.Handler):
    def emit(self, record):
        seen.append(record.getMessage())


logging.getLogger("zkp_subnet_amd.client").addHandler(Grab())
c = Client(seed=5)
c.start(10, 2)
c.stop()
out["client_warned_about_lanes"] = any("lanes run concurrently" in m for m in seen)
print(json.dumps(out), flush=True)
