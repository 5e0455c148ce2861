"""Tiered selection (csrc/nts_tiers.inc, nts_sketch_tiers) == every k-mer probed == CPU oracle, bit-exact, through the C ABI.

The regime: a common filter that accepts a few per cent of a genome's k-mers (BASELINE's scaling config before its last levels, the
reference's eleven-genome row, README.md:158) -- filter-in semantics as `indexlr -s` (bin/ntsynt_run_pipeline.smk:81-85).  Ragged
records (empty, shorter than k, shorter than w, exactly w k-mers), N runs, soft masks of a refinement round, records that start
inside a tile, tiles whose halo reaches into the neighbours, conserved stretches (every k-mer accepted) next to stretches that share
nothing.  Needs an MI355X."""
import numpy as np
import pytest

from oracle import nts_oracle as O
from tests.helpers import oracle_flat, random_records, to_device, to_oracle

pytestmark = pytest.mark.gpu

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


@pytest.fixture(scope="module")
def ctx():
    from ntsynt_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def _relative(rng, seqs, rate, conserved=(), novel=()):
    "the same records with substitutions at `rate`; `conserved` / `novel`: (record, start, end) kept as they are / replaced"
    out = []
    for r, s in enumerate(seqs):
        a = np.frombuffer(s, dtype=np.uint8).copy()
        keep = a.copy()
        hit = (rng.random(a.size) < rate) & (a != ord("N"))
        a[hit] = ACGT[rng.integers(0, 4, size=int(hit.sum()))]
        for rr, s0, s1 in conserved:
            if rr == r:
                a[s0:s1] = keep[s0:s1]
        for rr, s0, s1 in novel:
            if rr == r:
                a[s0:s1] = ACGT[rng.integers(0, 4, size=len(a[s0:s1]))]
        out.append(a.tobytes())
    return out


def _case(ctx, seed, lengths, k, fpr, rate, n_rel=2, **kw):
    from ntsynt_amd.device import BloomFilter
    rng = np.random.default_rng(seed)
    seqs = random_records(rng, lengths, n_frac=0.002)
    names = [f"r{i}" for i in range(len(seqs))]
    fam = [seqs] + [_relative(rng, seqs, rate, **kw) for _ in range(n_rel)]
    og = [to_oracle(names, s) for s in fam]
    dg = [to_device(ctx, names, s) for s in fam]
    nbytes = O.bf_ctor_bytes(O.bf_approx_bytes(og[0].total_bp, fpr))
    obf = None
    for o in og:
        obf = O.bf_build(o, k, nbytes, prev=obf)
    dbf = BloomFilter(ctx, nbytes, k)
    dbf.from_numpy(obf)
    return og, dg, obf, dbf


def _check(ctx, og, dg, obf, dbf, k, w, masks=None, **tiers):
    from ntsynt_amd.device import sketch
    n_total = 0
    for o, d in zip(og, dg):
        ctx.sketch_tiers("always", **tiers)
        got = sketch(ctx, d, k, w, dbf, masks).to_numpy()
        probes, rounds, n_tiers = ctx.sketch_tiers()
        assert n_tiers >= 2, "the call did not go through k_hash_tiers"
        assert probes > 0 and rounds > 0
        ctx.sketch_tiers("never")
        ctx.sketch_mode("dense")
        dense = sketch(ctx, d, k, w, dbf, masks).to_numpy()
        ctx.sketch_mode("auto")
        ctx.sketch_tiers("auto")
        for a, b in zip(got, dense):
            assert np.array_equal(a, b)
        if masks is None:
            exp = oracle_flat(O.minimize(o, k, w, obf))
            for a, b in zip(got, exp):
                assert np.array_equal(a, b.astype(a.dtype))
        # fewer probes than k-mers, or the tiers are pointless (tiny inputs aside)
        n_total += len(got[0])
    assert n_total > 0


LENGTHS = [150000, 0, 5, 23, 24, 25, 1023, 1046, 1047, 1500, 70001, 3, 12000, 40000, 16384 + 23, 900]


@pytest.mark.parametrize("k,w,rate,fpr", [(24, 1000, 0.10, 0.025), (24, 1000, 0.06, 0.025), (24, 250, 0.08, 0.025), (40, 1000, 0.05, 0.025),
                                          (24, 2000, 0.10, 0.025), (70, 500, 0.03, 0.025), (24, 64, 0.05, 0.3), (129, 300, 0.02, 0.025)])
def test_tiers_equal_dense_and_oracle(ctx, k, w, rate, fpr):
    if k > 128:
        pytest.skip("k_hash_tiers rolls k <= 128 (FAST_K_MAX); longer k-mers stay on the dense path")
    og, dg, obf, dbf = _case(ctx, 900 + k + w, LENGTHS, k, fpr, rate,
                             conserved=[(0, 20000, 26000), (10, 0, 3000)], novel=[(0, 60000, 75000), (13, 1000, 9000)])
    _check(ctx, og, dg, obf, dbf, k, w)


@pytest.mark.parametrize("k,w,rate", [(24, 10, 0.01), (24, 8, 0.002), (24, 16, 0.01), (24, 24, 0.01), (24, 25, 0.02), (24, 33, 0.02), (24, 63, 0.04), (16, 12, 0.01),
                                      (40, 20, 0.005), (96, 48, 0.003)])
def test_tiers_below_64_equal_the_window_tiles_and_the_oracle(ctx, k, w, rate):
    """Short windows (the last refinement round's w = 10, bin/ntSynt:89-91): probes in increasing hash order where a window is still open
    against k_window_min<true>, which probes every k-mer, and against the oracle.  With w <= k every substitution leaves a stretch of at
    least w k-mers without an accepted one: such windows have no minimizer on either path."""
    og, dg, obf, dbf = _case(ctx, 1200 + k + w, LENGTHS, k, 0.025, rate, conserved=[(0, 20000, 26000), (10, 0, 3000)], novel=[(0, 60000, 75000), (13, 1000, 9000)])
    _check(ctx, og, dg, obf, dbf, k, w)


def test_short_windows_go_through_the_tiers_by_themselves(ctx):
    "the automatic choice: tiers below w = 64 where their estimated probes pay (ntsynt_hip.hip: `pays`), the window tiles where they do not"
    from ntsynt_amd.device import sketch
    k = 24
    og, dg, obf, dbf = _case(ctx, 58, [300000, 2000, 90000], k, 0.025, 0.004, n_rel=1)
    for w, tiered in ((10, True), (33, True), (7, False)):
        mx = sketch(ctx, dg[0], k, w, dbf)
        got = mx.to_numpy()
        assert (ctx.sketch_tiers()[2] >= 2) == tiered, w
        exp = oracle_flat(O.minimize(og[0], k, w, obf))
        for a, b in zip(got, exp):
            assert np.array_equal(a, b.astype(a.dtype))
    # a filter that accepts a few per cent: nearly every k-mer would be probed anyway -- every k-mer is probed inside the window tiles
    og, dg, obf, dbf = _case(ctx, 59, [300000], k, 0.025, 0.08, n_rel=2)
    mx = sketch(ctx, dg[0], k, 33, dbf)
    assert ctx.sketch_tiers()[2] == 0
    exp = oracle_flat(O.minimize(og[0], k, 33, obf))
    for a, b in zip(mx.to_numpy(), exp):
        assert np.array_equal(a, b.astype(a.dtype))


@pytest.mark.parametrize("k,w", [(24, 10), (24, 33), (24, 63), (24, 70), (16, 8), (64, 25)])
def test_short_windows_without_a_filter_go_through_the_tiers(ctx, k, w):
    """ntSynt --no-common (indexlr without -s, bin/ntsynt_run_pipeline.smk:81-85) below the window where one threshold takes over: every listed
    k-mer is accepted, the rounds decide which k-mers are hashed in full at all.  Against the window tiles and the oracle."""
    from ntsynt_amd.device import sketch
    rng = np.random.default_rng(300 + k + w)
    seqs = random_records(rng, LENGTHS, n_frac=0.002)
    names = [f"r{i}" for i in range(len(seqs))]
    o, d = to_oracle(names, seqs), to_device(ctx, names, seqs)
    got = sketch(ctx, d, k, w, None).to_numpy()
    assert ctx.sketch_tiers()[2] >= 2, "the call did not go through k_hash_tiers"
    ctx.sketch_mode("dense")
    dense = sketch(ctx, d, k, w, None).to_numpy()
    ctx.sketch_mode("auto")
    exp = oracle_flat(O.minimize(o, k, w, None))
    for a, b, c in zip(got, dense, exp):
        assert np.array_equal(a, b) and np.array_equal(a, c.astype(a.dtype))


@pytest.mark.parametrize("x0,half", [(0.2, False), (0.7, True), (2.4, True), (6.0, False), (8.0, False)])
def test_tier_schedules_give_the_same_list(ctx, x0, half):
    "many thin tiers, few fat ones, steps of 1.5: the schedule changes the probes, never the result"
    k, w = 24, 1000
    og, dg, obf, dbf = _case(ctx, 77, LENGTHS, k, 0.025, 0.09, conserved=[(0, 100000, 103000)], novel=[(10, 30000, 50000)])
    _check(ctx, og, dg, obf, dbf, k, w, x0=x0, half_steps=half)


def test_tiers_with_masks_of_a_refinement_round(ctx):
    k, w = 24, 250
    og, dg, obf, dbf = _case(ctx, 31, LENGTHS, k, 0.025, 0.07)
    masks = [(0, 1000, 30000), (0, 90000, 90100), (10, 100, 50000), (13, 0, 40000), (14, 16000, 16384 + 23)]
    _check(ctx, og, dg, obf, dbf, k, w, masks=masks)


def test_tiers_on_many_short_records(ctx):
    "thousands of records per tile: record starts are what closes a stretch, almost everywhere"
    rng = np.random.default_rng(5)
    lengths = [int(x) for x in rng.integers(0, 2500, size=400)] + [60000]
    k, w = 24, 500
    og, dg, obf, dbf = _case(ctx, 6, lengths, k, 0.025, 0.08)
    _check(ctx, og, dg, obf, dbf, k, w)


def test_tiers_probe_a_fraction_of_the_kmers(ctx):
    "at an accepted share of a few per cent the rounds probe a small part of the k-mers, not all of them"
    from ntsynt_amd.device import sketch
    k, w = 24, 1000
    og, dg, obf, dbf = _case(ctx, 41, [400000], k, 0.025, 0.10, n_rel=2)
    ctx.sketch_tiers("always")
    mx = sketch(ctx, dg[0], k, w, dbf)
    probes, rounds, n_tiers = ctx.sketch_tiers()
    ctx.sketch_tiers("auto")
    n_kmers = dg[0].valid_kmers(k)
    assert len(mx) > 0 and n_tiers >= 3
    assert probes < 0.45 * n_kmers, (probes, n_kmers)


@pytest.mark.parametrize("n,div,w", [(3, 0.10, 1000), (4, 0.06, 500)])
def test_pipeline_in_the_valley_is_byte_identical_to_the_oracle_pipeline(ctx, tmp_path, n, div, w):
    """FASTA files of a family whose common filter accepts a few per cent of the k-mers -> the whole product pipeline (GPU parse, fused
    cascade, sketches through k_hash_tiers, graph stage in HBM, two refinement rounds) == the oracle
    pipeline: both synteny TSVs and every minimizer TSV byte for byte (`ntSynt -d 10`-like distances, bin/ntSynt:95-97)"""
    import os
    from ntsynt_amd import pipeline, synth
    from oracle import synteny_oracle as SO
    paths = synth.make_family(str(tmp_path), n, 2_500_000, 3, div, seed=77 + n, micro=4, n_runs=True, soft_mask=True)
    kw = dict(k=24, w=w, w_rounds=[100, 10], indel=10000, merge=10000, block_size=500, prefix="v")
    cwd = os.getcwd()
    ctx.profile(1)
    ctx.sketch_tiers("always")                  # (files of a few Mbp: the library would take the sparse-filter path, which applies to small filters too)
    try:
        os.makedirs(tmp_path / "hip")
        os.makedirs(tmp_path / "ora")
        os.chdir(tmp_path / "hip")
        try:
            eng = pipeline.run(paths, log=lambda *a: None, ctx=ctx, **kw)
            hip_out = eng.outputs
        except SystemExit:                      # "no paths found" (S:630-632): then on both sides
            hip_out = None
        tiers_ms, tiers_launches = ctx.timing("hash_tiers")
        os.chdir(tmp_path / "ora")
        try:
            ora_out = SO.run_pipeline(paths, **kw).outputs
        except SystemExit:
            ora_out = None
    finally:
        os.chdir(cwd)
        ctx.profile(0)
        ctx.sketch_tiers("auto")
    assert tiers_launches >= 1, "the whole-genome sketches did not go through the tiered selection"
    assert (hip_out is None) == (ora_out is None)
    if hip_out is not None:
        for name in ("v.synteny_blocks.tsv", "v.pre-collinear-merge.synteny_blocks.tsv"):
            assert hip_out[name] == ora_out[name], name
    for p in paths:
        name = f"{os.path.basename(p)}.k24.w{w}.tsv"
        assert open(tmp_path / "hip" / name).read() == open(tmp_path / "ora" / name).read(), name
