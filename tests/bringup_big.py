"""Bring-up aid (not collected by pytest): one large MSM, trapdoor-checked against the oracle.  python tests/bringup_big.py LOG2_N"""
import sys,time,os
sys.path.insert(0,'.')
import numpy as np
from oracle import bls12_381 as o, cpu as oc
from zkp_subnet_amd import HipEngine
lg=int(sys.argv[1]); n=1<<lg
eng=HipEngine(0)
tx=0x1234567890ABCDEF1234567
t=time.time(); eng.gen_srs(tx,1,lg,0); print("gen+tables",time.time()-t, "window",eng.window, flush=True)
raw=np.random.default_rng(5).integers(0,256,size=(n,32),dtype=np.uint8); raw[:,0]&=0x3F
scb=raw.tobytes(); del raw
eng.upload_fr(0,scb,False); eng.set_profiling(True)
t=time.time(); got=eng.msm_resident(0,n,0); print("msm first",time.time()-t)
t=time.time(); got2=eng.msm_resident(0,n,0); dt=time.time()-t; print("msm",dt, n/dt/1e6,"Mpts/s", eng.timings(), flush=True)
t=time.time(); y=oc.fr_eval(scb, tx.to_bytes(32,'big')); exp=oc.g1_mul_gen(y); print("oracle",time.time()-t)
print("trapdoor", got==exp and got2==exp)
