"""Timing of the native chain walk on a 6M-vertex graph of shuffled chains (NTS_HOST_DEBUG=1 prints the phases)."""
import numpy as np, time, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntsynt_amd.graph import walk_chains
rng=np.random.default_rng(1)
nv=6_000_000
perm=rng.permutation(nv)
cut=np.sort(rng.choice(nv-1, 4000, replace=False))
eu=perm[:-1]; ev=perm[1:]
keep=np.ones(nv-1,bool); keep[cut]=False
eu=eu[keep]; ev=ev[keep]
o=rng.permutation(eu.size); eu=eu[o]; ev=ev[o]
for i in range(2):
    t=time.time(); off,verts=walk_chains(nv,eu,ev); print(time.time()-t, off.size)
