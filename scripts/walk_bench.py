"""Timing of the native chain walk on a 6M-vertex graph of shuffled chains (NTS_HOST_DEBUG=1 prints the phases)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntsynt_amd.graph import edge_degrees, walk_paths  # noqa: E402

rng = np.random.default_rng(1)
nv = 6_000_000
perm = rng.permutation(nv)
cut = np.sort(rng.choice(nv - 1, 4000, replace=False))
eu, ev = perm[:-1], perm[1:]
keep = np.ones(nv - 1, bool)
keep[cut] = False
eu, ev = eu[keep], ev[keep]
o = rng.permutation(eu.size)
eu, ev = eu[o].astype(np.int64), ev[o].astype(np.int64)
alive = np.ones(eu.size, bool)
key = rng.permutation(nv).astype(np.int64)
for i in range(3):
    t = time.time()
    off, verts = walk_paths(nv, eu, ev, alive, key)
    t1 = time.time()
    deg = edge_degrees(nv, eu, ev, alive)
    print(f"walk_paths {t1 - t:.3f} s ({off.size - 1} paths), edge_degrees {time.time() - t1:.3f} s", flush=True)
