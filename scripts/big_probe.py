"""Genomes beyond 2^32 bp (one record crossing 2^32): pruned == dense == oracle on slices either side of the 2^32 boundary."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from oracle import nts_oracle as O
from ntsynt_amd.device import BloomFilter, Context, Genome, bf_size_bytes, sketch

ctx = Context(0)
k, w = 24, 1000
total, contigs = int(float(sys.argv[1]) * 1e9) if len(sys.argv) > 1 else 9_200_000_000, 2
g0 = Genome.synth(ctx, total, contigs, 77, 1, 0.005)
g1 = Genome.synth(ctx, total, contigs, 77, 2, 0.005)
print("total", g0.total_bp, "valid", g0.valid_kmers(k), total - contigs * (k - 1))
_, nbytes = bf_size_bytes(total, 0.025)
common = BloomFilter(ctx, nbytes, k)
common.insert(g0)
print("occ0", common.get_fpr())
other = BloomFilter(ctx, nbytes, k)
other.insert(g1)
common.and_(other)
other.free()
print("occ", common.get_fpr())
res = {}
for mode in ("pruned", "dense"):
    ctx.sketch_mode(mode)
    t = time.time()
    res[mode] = sketch(ctx, g1, k, w, common).to_numpy()
    print(mode, len(res[mode][0]), round(time.time() - t, 3))
for a, b in zip(res["pruned"], res["dense"]):
    assert np.array_equal(a, b)
h1, rec, pos = res["pruned"]
per = total // contigs
print("max pos", int(pos.max()), "per", per)
assert int(pos.max()) > (1 << 32)
bits = common.to_numpy()
n_slice = 1_000_000
for r in (0, 1):
    for start in (0, (1 << 32) - 500_000, per - n_slice):
        seq = g1.download(int(g1.rec_off[r]) + start, n_slice).tobytes()
        exp = O.minimize(O.Genome(["s"], [seq]), k, w, bits)[0]
        # windows wholly inside the slice: compare the interior
        m = (rec == r) & (pos >= start + w + k) & (pos < start + n_slice - k - w)
        e = (exp[1] >= w + k) & (exp[1] < n_slice - k - w)
        assert m.sum() > 500, m.sum()
        assert np.array_equal(pos[m] - start, exp[1][e].astype(np.uint64)) and np.array_equal(h1[m], exp[0][e]), (r, start)
        print("slice ok", r, start, int(m.sum()))
print("OK")
