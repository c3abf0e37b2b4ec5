import sys; sys.path.insert(0,'.')
from oracle import bls12_381 as o
from zkp_subnet_amd import HipEngine
e=HipEngine(0)
def show(v,inv):
    out=e.ntt(o.fr_to_be32(v),inv); got=o.fr_from_be32(out); exp=o.ntt(v,inv)
    print(v if max(v)<100 else '...',inv,[hex(x) for x in got], 'exp',[hex(x) for x in exp], got==exp)
    if got!=exp and len(v)>=2:
        # ratio got/exp
        for g,x in zip(got,exp):
            if x: print('  ratio',hex(g*pow(x,-1,o.R)%o.R))
show([1,1],True); show([1,0],True); show([3,5],True); show([3,5],False); show([1,2,3,4],True); show([1,2,3,4],False)
