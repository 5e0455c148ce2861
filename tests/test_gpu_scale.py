"""Human-scale inputs (BASELINE.json configs[2] shape: 3 Gbp per genome, k=24 w=1000) generated directly in
HBM, checked through size-independent properties and oracle spot-checks on slices read back from the device."""
import numpy as np
import pytest

from oracle import nts_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from ntsynt_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def test_three_gbp_genome_sketch_properties_and_slice_parity(ctx):
    from ntsynt_amd.device import BloomFilter, Genome, bf_size_bytes, sketch
    k, w, contigs, total = 24, 1000, 24, 3_000_000_000
    g0 = Genome.synth(ctx, total, contigs, 20240207, 1, 0.005)
    g1 = Genome.synth(ctx, total, contigs, 20240207, 2, 0.005)
    assert g0.total_bp == total and g0.valid_kmers(k) == total - contigs * (k - 1)
    approx, nbytes = bf_size_bytes(total, 0.025)
    assert approx == O.bf_approx_bytes(total, 0.025) == 14811708827
    common = BloomFilter(ctx, nbytes, k)
    common.insert(g0)
    occ0 = common.get_fpr()
    assert abs(occ0 - 0.025) < 0.001                # ~ -expm1(-n/m): the sizing rule's target occupancy
    # the partitioned build (two bucket levels at this size) and one atomic per k-mer set the same 14.8 GB of bits
    ctx.bf_build_mode("atomic")
    atomic = BloomFilter(ctx, nbytes, k)
    atomic.insert(g0)
    ctx.bf_build_mode("auto")
    n_bits = common.popcount()
    assert atomic.popcount() == n_bits
    atomic.and_(common)
    assert atomic.popcount() == n_bits              # same count and a full intersection: the same set
    atomic.free()
    other = BloomFilter(ctx, nbytes, k)
    other.insert(g1)
    common.and_(other)
    other.free()
    occ = common.get_fpr()
    assert 0.015 < occ < occ0
    res = {}
    for mode in ("pruned", "dense"):
        ctx.sketch_mode(mode)
        res[mode] = sketch(ctx, g1, k, w, common).to_numpy()
    ctx.sketch_mode("auto")
    h1, rec, pos = res["pruned"]
    for a, b in zip(res["pruned"], res["dense"]):       # pruning is exact
        assert np.array_equal(a, b)
    assert 0.9 * 2 * total / (w + 1) < h1.size < 1.3 * 2 * total / (w + 1)
    per = total // contigs
    for r in (0, 7, contigs - 1):
        p = pos[rec == r].astype(np.int64)
        assert (np.diff(p) > 0).all() and p[-1] <= per - k
    # oracle on slices read back from HBM, with the device-built filter
    bits = common.to_numpy()
    n_slice = 1_500_000
    for r in (0, contigs - 1):
        seq = g1.download(int(g1.rec_off[r]), n_slice).tobytes()
        exp = O.minimize(O.Genome(["s"], [seq]), k, w, bits)[0]
        m = (rec == r) & (pos < n_slice - k - w)
        n = int(m.sum())
        assert n > 1000
        assert np.array_equal(pos[m], exp[1][:n]) and np.array_equal(h1[m], exp[0][:n])
    # spot-check of the filter itself: every k-mer of a slice of g1 that g0 also holds must be present
    seq0 = g0.download(int(g0.rec_off[3]), 200_000).tobytes()
    seq1 = g1.download(int(g1.rec_off[3]), 200_000).tobytes()
    p0, h0a = O.hash_all(seq0, k)
    p1, h0b = O.hash_all(seq1, k)
    shared = np.intersect1d(h0a, h0b)
    assert shared.size > 100_000
    for h in shared[::997]:
        assert O.bf_contains(bits, int(h))
    for g in (g0, g1):
        g.free()
    common.free()


def test_records_longer_than_2_to_32(ctx):
    """Two 4.6 Gbp records (lungfish-scale chromosomes; 46 GB filter): pruned == dense, positions beyond 2^32 come back, and the
    oracle agrees on slices either side of the 2^32 boundary of both records."""
    from ntsynt_amd.device import BloomFilter, Genome, bf_size_bytes, sketch
    k, w, contigs, total = 24, 1000, 2, 9_200_000_000
    g0 = Genome.synth(ctx, total, contigs, 77, 1, 0.005)
    g1 = Genome.synth(ctx, total, contigs, 77, 2, 0.005)
    assert g0.total_bp == total and g0.valid_kmers(k) == total - contigs * (k - 1)
    _, nbytes = bf_size_bytes(total, 0.025)
    common = BloomFilter(ctx, nbytes, k)
    common.insert(g0)
    assert abs(common.get_fpr() - 0.025) < 0.001
    other = BloomFilter(ctx, nbytes, k)
    other.insert(g1)
    common.and_(other)
    other.free()
    res = {}
    for mode in ("pruned", "dense"):
        ctx.sketch_mode(mode)
        res[mode] = sketch(ctx, g1, k, w, common).to_numpy()
    ctx.sketch_mode("auto")
    for a, b in zip(res["pruned"], res["dense"]):
        assert np.array_equal(a, b)
    h1, rec, pos = res["pruned"]
    per = total // contigs
    assert int(pos.max()) > (1 << 32) and int(pos.max()) <= per - k
    bits = common.to_numpy()
    n_slice = 1_000_000
    for r in (0, 1):
        for start in (0, (1 << 32) - 500_000, per - n_slice):
            seq = g1.download(int(g1.rec_off[r]) + start, n_slice).tobytes()
            exp = O.minimize(O.Genome(["s"], [seq]), k, w, bits)[0]
            m = (rec == r) & (pos >= start + w + k) & (pos < start + n_slice - k - w)   # windows wholly inside the slice
            e = (exp[1] >= w + k) & (exp[1] < n_slice - k - w)
            assert int(m.sum()) > 500
            assert np.array_equal(pos[m] - np.uint64(start), exp[1][e].astype(np.uint64)) and np.array_equal(h1[m], exp[0][e])
    for g in (g0, g1):
        g.free()
    common.free()


@pytest.mark.parametrize("record,family,div,n_fam,mbp,contigs", [
    ("r04_e2e_oracle_c3.json", "structural", 0.01, 3, 3000, 24), ("r04_e2e_oracle_c5_like.json", "assembly-like", 0.013, 3, 3000, 24),
    ("r04_e2e_oracle_2x3Gbp_d0.1.json", "structural", 0.001, 2, 3000, 24), ("r04_e2e_oracle_c4.json", "structural", 0.10, 8, 3000, 24),
    # (round 5) three genomes at 10 %: a filter that accepts 2.4 % of the k-mers, the tiered selection's regime
    ("r05_e2e_oracle_valley_3x10pct.json", "structural", 0.10, 3, 3000, 24),
    # (round 6) the reference's other two published rows (README.md:157-158) as shapes: four 3 Gbp assembly-like genomes at 1.3 %
    # ("great apes"), eleven 0.44 Gbp genomes of 16 chromosomes at 4 % ("bees")
    ("r06_e2e_oracle_4x3Gbp_apes_like.json", "assembly-like", 0.013, 4, 3000, 24), ("r06_e2e_oracle_11x440Mbp_4pct.json", "structural", 0.04, 11, 440, 16)])
def test_whole_output_at_headline_size_equals_the_recorded_oracle_run(ctx, record, family, div, n_fam, mbp, contigs):
    """BASELINE configs[2] (and configs[4]'s parameters on the assembly-like family, and the reference's first published row: two 3 Gbp
    genomes at 0.1 %, README.md:156, and configs[3]: eight genomes at 10 %, the sparse-filter regime) at full size, the WHOLE output: the common
    filter's popcount and an order-independent digest of every genome's complete minimizer list (3 x ~6 M minimizers) against what
    the CPU oracle pipeline left on record when it ran on these very families on a GPU box's host cores (scripts/e2e_oracle_check.py
    -> profiles/: 8 minutes of CPU, so it is a record, not a step of the suite).  The slices above compare a few Mbp with a live
    oracle; this compares everything with a recorded one."""
    import argparse
    import json
    import os
    import bench
    from ntsynt_amd.device import BloomFilter, bf_size_bytes, sketch
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", record)
    if not os.path.exists(path):
        pytest.skip(f"{record}: no full-size oracle record committed")
    rec = json.load(open(path))
    assert rec["all_identical"]
    args = argparse.Namespace(family=family, substitutions_only=False, k=24, w=1000, fpr=0.025)
    total = int(mbp * 1e6)
    assert rec["key"] == bench.e2e_key(args, n_fam, total, contigs, div), "the record is for another family / parameter set"
    genomes = [bench.family_genome(ctx, args, total, contigs, j, div / 2.0) for j in range(n_fam)]
    _, nbytes = bf_size_bytes(genomes[0].total_bp, 0.025)           # (syn0.fa sorts first: it sizes the filter, cpp:105-118)
    assert nbytes == rec["oracle_filter_bytes"]
    common = BloomFilter(ctx, nbytes, 24)
    common.insert(genomes[0])
    for g in genomes[1:]:
        common.insert_and(g)
    assert common.popcount() == rec["oracle_filter_popcount"]
    for j, g in enumerate(genomes):
        mx = sketch(ctx, g, 24, 1000, common)
        h1, r, pos = mx.to_numpy()
        mx.free()
        assert bench.mx_digest(h1, r, pos) == rec["oracle_minimizer_digests"][f"syn{j}.fa.k24.w1000.tsv"], j
    common.free()
    for g in genomes:
        g.free()
