import sys, json, subprocess
sys.path.insert(0, '.')
from zkp_subnet_amd.engine import HipEngine
from bench import uniform_fr, TAU
outs = {}
for c in (20, 22, 24):
    eng = HipEngine(0, window=c)
    eng.gen_srs(TAU, 1, 22, 0)
    eng.upload_fr(0, uniform_fr(1 << 22, seed=5), False)
    outs[c] = eng.msm_resident(0, 1 << 22, 0)
    print(c, eng.msm_plan(1 << 22), outs[c].hex()[:32], flush=True)
    del eng
assert outs[20] == outs[22] == outs[24], "window 24 disagrees"
print("window 24 == 22 == 20 at 2^22")
