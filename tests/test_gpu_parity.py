"""HIP path vs CPU oracle, bit-exact, through the C ABI (rows A1-A4, B1-B3, B5).  Needs an MI355X."""
import numpy as np
import pytest

from oracle import nts_oracle as O
from tests.helpers import oracle_flat, random_records, to_device, to_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from ntsynt_amd.device import Context
    c = Context(0)
    yield c
    c.close()


LENGTHS = [30000, 0, 5, 23, 24, 25, 1023, 1024, 1500, 70001, 3, 12000]


def _family(seed, lengths=LENGTHS, **kw):
    rng = np.random.default_rng(seed)
    seqs = random_records(rng, lengths, **kw)
    names = [f"r{i}" for i in range(len(seqs))]
    return names, seqs


@pytest.mark.parametrize("k", [20, 24, 31, 64, 128, 129, 200])   # above 128 every kernel walks the run table (no LDS staging)
def test_hash_all(ctx, k):
    names, seqs = _family(10 + k)
    dev = to_device(ctx, names, seqs)
    got = dev.hash_all(k)
    exp = np.concatenate([O.hash_all(s, k)[1] for s in seqs])
    assert dev.valid_kmers(k) == exp.size
    assert np.array_equal(got, exp)


def test_bf_size_matches_oracle():
    from ntsynt_amd.device import bf_size_bytes
    for n in (1, 1000, 29058289, 10**8, 3 * 10**9):
        for fpr in (0.025, 0.01, 0.3):
            a, c = bf_size_bytes(n, fpr)
            assert a == O.bf_approx_bytes(n, fpr)
            assert c == O.bf_ctor_bytes(a)


@pytest.mark.parametrize("k", [20, 24])
def test_bloom_insert_cascade_and(ctx, k):
    from ntsynt_amd.device import BloomFilter
    fams = [_family(100 + i, lengths=[40000, 0, 9000, 17], n_frac=0.005) for i in range(3)]
    # make them related so the common filter is not empty
    base = fams[0][1]
    rng = np.random.default_rng(5)
    seqs_list = [base]
    for j in (1, 2):
        mut = []
        for s in base:
            a = np.frombuffer(s, dtype=np.uint8).copy()
            hit = rng.random(a.size) < 0.01
            a[hit] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(hit.sum()))]
            mut.append(a.tobytes())
        seqs_list.append(mut)
    names = fams[0][0]
    og = [to_oracle(names, s) for s in seqs_list]
    dg = [to_device(ctx, names, s) for s in seqs_list]
    nbytes = O.bf_ctor_bytes(O.bf_approx_bytes(og[0].total_bp, 0.025))
    # level 1
    o1 = O.bf_build(og[0], k, nbytes)
    d1 = BloomFilter(ctx, nbytes, k)
    d1.insert(dg[0])
    assert np.array_equal(d1.to_numpy(), o1)
    assert d1.popcount() == O.bf_popcount(o1)
    # literal cascade (cpp:134-160)
    o_prev, d_prev = o1, d1
    for j in (1, 2):
        o_next = O.bf_build(og[j], k, nbytes, prev=o_prev)
        d_next = BloomFilter(ctx, nbytes, k)
        d_next.cascade_from(d_prev, dg[j])
        assert np.array_equal(d_next.to_numpy(), o_next)
        o_prev, d_prev = o_next, d_next
    # AND of independent per-genome filters == cascade
    acc = BloomFilter(ctx, nbytes, k)
    acc.insert(dg[0])
    for j in (1, 2):
        gj = BloomFilter(ctx, nbytes, k)
        gj.insert(dg[j])
        acc.and_(gj)
    assert np.array_equal(acc.to_numpy(), o_prev)
    assert abs(acc.get_fpr() - O.bf_fpr(o_prev)) < 1e-15
    # upload / download round trip
    rt = BloomFilter(ctx, nbytes, k)
    rt.from_numpy(o_prev)
    assert rt.popcount() == O.bf_popcount(o_prev)


@pytest.mark.parametrize("nbytes", [3 << 20, 100 << 20])   # 24 final buckets (one partition level) / 800 (two levels)
def test_bloom_binned_build_equals_atomic_and_oracle(ctx_x, nbytes, monkeypatch):
    """nts_bf_insert's partitioned build (hash -> bucket passes -> LDS bitmaps) sets the same bits as one atomic OR
    per k-mer and as the oracle: random records with N runs (tiles that cross run boundaries take the direct path),
    a repeat that piles 300k copies of a handful of k-mers into a few buckets (capacity overflow -> direct atomics),
    and inserting into a filter that already holds bits."""
    ctx = ctx_x            # (environment switches of the experiments build: tests/conftest.py)
    from ntsynt_amd.device import BloomFilter
    k = 24
    names, seqs = _family(77, lengths=[400000, 0, 30, 250000, 12000, 90001], n_frac=0.0002)
    seqs = list(seqs) + [b"ACGTTGCA" * 40000, b"A" * 300000]
    names = [f"r{i}" for i in range(len(seqs))]
    og, dg = to_oracle(names, seqs), to_device(ctx, names, seqs)
    names2, seqs2 = _family(78, lengths=[150000, 70000])
    og2, dg2 = to_oracle(names2, seqs2), to_device(ctx, names2, seqs2)
    want = O.bf_build(og, k, nbytes)
    want2 = want | O.bf_build(og2, k, nbytes)
    got = {}
    try:
        # binned into an empty filter: bitmaps stored without reading the filter, indices that bypass the buckets parked in a list
        # and set afterwards; "read+or": the same build made to read and OR (NTS_BIN_STORE=0); "list full": the parking list
        # cut to 1000 entries, so that it runs over and the finish falls back to read-and-OR on the device's own say
        for mode in ("atomic", "binned", "binned read+or", "binned list full"):
            ctx.bf_build_mode(mode.split()[0])
            monkeypatch.delenv("NTS_BIN_STORE", raising=False)
            monkeypatch.delenv("NTS_BIN_LATE_CAP", raising=False)
            if mode == "binned read+or":
                monkeypatch.setenv("NTS_BIN_STORE", "0")
            if mode == "binned list full":
                monkeypatch.setenv("NTS_BIN_LATE_CAP", "1000")
            bf = BloomFilter(ctx, nbytes, k)
            bf.insert(dg)
            got[mode] = bf.to_numpy()
            assert bf.popcount() == int(np.unpackbits(want).sum())
            bf.insert(dg2)                       # OR into a non-empty filter
            assert np.array_equal(bf.to_numpy(), want2), mode
            bf.clear()                           # a cleared filter counts as empty again
            bf.insert(dg2)
            assert np.array_equal(bf.to_numpy(), O.bf_build(og2, k, nbytes)), mode
            bf.free()
    finally:
        ctx.bf_build_mode("auto")
    for mode, bits in got.items():
        assert np.array_equal(bits, want), mode



@pytest.mark.parametrize("nbytes", [3 << 20, 100 << 20])   # 24 final buckets (one partition level) / 800 (two levels)
def test_bloom_fused_and_equals_insert_then_and_and_the_oracle_cascade(ctx_x, nbytes, monkeypatch):
    """nts_bf_insert_and (the cascade level of cpp:134-160 as acc &= G inside the partitioned build's last pass) against insert
    into a second filter + nts_bf_and, against the literal cascade kernel and against the oracle's cascade -- on records whose
    repeats overflow their buckets (the copies of a k-mer park ONE index and the bit comes back after the AND), with tiles in
    pieces, with the parking list cut short (the device gives up, restores the running filter, the level is redone the plain
    way), on a running filter so sparse that slices are skipped, and on all-ones / all-zero running filters."""
    ctx = ctx_x            # (environment switches of the experiments build: tests/conftest.py)
    from ntsynt_amd.device import BloomFilter
    k = 24
    rng = np.random.default_rng(2024)
    names, seqs = _family(81, lengths=[400000, 0, 30, 250000, 12000, 90001], n_frac=0.0002)
    rep = [b"ACGTTGCA" * 40000, b"A" * 300000]
    seqs_a = list(seqs) + rep
    # a relative: the same records at 2 % divergence (+ the same repeats, so that overflowing bits ARE in the running filter)
    seqs_b = []
    for s_ in seqs:
        a = np.frombuffer(s_, dtype=np.uint8).copy()
        hit = (rng.random(a.size) < 0.02) & (a != ord("N"))
        a[hit] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(hit.sum()))]
        seqs_b.append(a.tobytes())
    seqs_b += rep
    seqs_c = [seqs_b[0][:200000], b"ACGTTGCA" * 20000]        # a third genome: little in common, the repeat again
    fam = []
    for sq in (seqs_a, seqs_b, seqs_c):
        nm = [f"r{i}" for i in range(len(sq))]
        fam.append((to_oracle(nm, sq), to_device(ctx, nm, sq)))
    o = O.bf_build(fam[0][0], k, nbytes)
    levels = [o]
    for og, _ in fam[1:]:
        o = O.bf_build(og, k, nbytes, prev=o)
        levels.append(o)
    assert 0 < int(np.unpackbits(levels[2]).sum()) < int(np.unpackbits(levels[1]).sum()) < int(np.unpackbits(levels[0]).sum())
    try:
        for mode in ("binned", "binned list full", "binned unfused", "atomic", "auto"):
            ctx.bf_build_mode(mode.split()[0])
            monkeypatch.delenv("NTS_BIN_LATE_CAP", raising=False)
            monkeypatch.delenv("NTS_BIN_FUSED_AND", raising=False)
            if mode == "binned list full":
                monkeypatch.setenv("NTS_BIN_LATE_CAP", "3")
            if mode == "binned unfused":
                monkeypatch.setenv("NTS_BIN_FUSED_AND", "0")
            acc = BloomFilter(ctx, nbytes, k)
            acc.insert(fam[0][1])
            for lvl, (_, dg) in enumerate(fam[1:], start=1):
                if lvl == 2:
                    acc.popcount()               # the library now knows how sparse the running filter is (slice-skipping path)
                acc.insert_and(dg)
                assert np.array_equal(acc.to_numpy(), levels[lvl]), (mode, lvl)
                st = ctx.path_stats()
                if mode == "binned":
                    assert st["bf_direct_indices"] > 0 and st["bf_list_fallback"] == 0, st
                if mode == "binned list full" and lvl == 1:
                    assert st["bf_list_fallback"] == 1, st
            assert acc.popcount() == int(np.unpackbits(levels[2]).sum())
            # identities of AND: all ones -> the genome's own filter; all zeros stay zeros
            ones = BloomFilter(ctx, nbytes, k, ones=True)
            ones.insert_and(fam[1][1])
            assert np.array_equal(ones.to_numpy(), O.bf_build(fam[1][0], k, nbytes)), mode
            zero = BloomFilter(ctx, nbytes, k)
            zero.insert_and(fam[1][1])
            assert zero.popcount() == 0, mode
            for f in (acc, ones, zero):
                f.free()
    finally:
        ctx.bf_build_mode("auto")


def _diverged(rng, seqs, d):
    out = []
    for s_ in seqs:
        a = np.frombuffer(s_, dtype=np.uint8).copy()
        hit = (rng.random(a.size) < d) & (a != ord("N"))
        a[hit] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(hit.sum()))]
        out.append(a.tobytes())
    return out


@pytest.mark.parametrize("variant", ["auto", "LDS-staged accept kernel", "summary only", "forced from level 1", "k=40", "repeats"])
def test_bloom_sparse_level_equals_the_build_and_the_oracle_cascade(ctx_x, variant, monkeypatch):
    """A cascade level over a running filter that is all but empty goes the reference's literal way (cpp:134-160: every k-mer of
    the genome looked up, the bits that were hit kept -- bf_level_sparse) instead of through a whole partitioned build: same bits as
    the build with the AND in its last pass, as the oracle's cascade, same popcount, through each of the three accept kernels, from
    the first level on when forced, with k > 32, and with repeats whose copies overflow the accept lists (the level then falls
    back to the build, the running filter untouched).  The summary and folded tables the level leaves behind are the ones the
    sketch uses next: minimizers with the final filter == oracle."""
    ctx = ctx_x            # (environment switches of the experiments build: tests/conftest.py)
    from ntsynt_amd.device import BloomFilter, sketch
    k = 40 if variant == "k=40" else 24
    w = 50
    nbytes = 100 << 20
    rng = np.random.default_rng(4242)
    names, anc = _family(91, lengths=[400000, 0, 30, 250000, 12000, 90001, 23, 24], n_frac=0.0002)
    rep = [b"ACGTTGCA" * 40000, b"A" * 300000] if variant == "repeats" else []
    fam = []
    for j in range(5):
        sq = _diverged(rng, anc, 0.03) + rep
        nm = [f"r{i}" for i in range(len(sq))]
        fam.append((to_oracle(nm, sq), to_device(ctx, nm, sq)))
    o = O.bf_build(fam[0][0], k, nbytes)
    levels = [o]
    for og, _ in fam[1:]:
        o = O.bf_build(og, k, nbytes, prev=o)
        levels.append(o)
    pops = [int(np.unpackbits(x).sum()) for x in levels]
    assert pops[0] > pops[1] > pops[2] > pops[3] > pops[4] > 0, pops
    if variant == "LDS-staged accept kernel":
        monkeypatch.setenv("NTS_ACCEPT_REG", "0")
    if variant == "forced from level 1":
        monkeypatch.setenv("NTS_BF_SPARSE_MAX_OCC", "1.0")
    if variant == "summary only":                                           # (not the library's own choice: never ahead of the build)
        monkeypatch.setenv("NTS_BF_SPARSE_MAX_OCC", "0.0004")
    try:
        if variant == "summary only":
            ctx.sketch_summary("no-lds")
        res = {}
        for how in ("sparse", "build"):
            if how == "build":
                monkeypatch.setenv("NTS_BF_SPARSE_LEVEL", "0")
            acc = BloomFilter(ctx, nbytes, k)
            acc.insert(fam[0][1])
            went = []
            for lvl, (_, dg) in enumerate(fam[1:], start=1):
                if variant == "forced from level 1":
                    acc.popcount()                                          # (the library must know the popcount to consider the literal level)
                acc.insert_and(dg)
                st = ctx.bf_level_stats()
                went.append(st["sparse_level"])
                assert acc.popcount() == pops[lvl], (variant, how, lvl, st)
                assert np.array_equal(acc.to_numpy(), levels[lvl]), (variant, how, lvl, st)
                if st["sparse_level"]:
                    assert st["accepted_kmers"] >= pops[lvl], st          # every kept bit was hit by at least one k-mer
            if how == "build":
                assert went == [0, 0, 0, 0], went
            elif variant == "forced from level 1":
                assert went == [1, 1, 1, 1], went
            elif variant == "repeats":
                assert went[0] == 0, went                                   # (what the later levels did depends on how the lists fell)
            else:
                assert went[0] == 0 and went[-1] == 1 and sum(went) >= 2, (went, pops)   # the library chose it once the filter was sparse
            # the tables the last level left with the filter serve the sketch
            for og, dg in fam[:2]:
                h1, rec, pos = sketch(ctx, dg, k, w, acc).to_numpy()
                exp = O.minimize(og, k, w, levels[-1])
                assert np.array_equal(h1, np.concatenate([e[0] for e in exp])) and np.array_equal(pos, np.concatenate([e[1] for e in exp])), (variant, how)
            res[how] = went
            acc.free()
    finally:
        ctx.sketch_summary("auto")
        for _, dg in fam:
            dg.free()


@pytest.mark.parametrize("k,w", [(24, 1000), (24, 100), (20, 10), (24, 1), (32, 17), (24, 16), (24, 15),
                                 (24, 4097), (20, 250)])
def test_sketch_no_filter(ctx, k, w):
    from ntsynt_amd.device import sketch
    names, seqs = _family(1000 + w)
    og, dg = to_oracle(names, seqs), to_device(ctx, names, seqs)
    exp = oracle_flat(O.minimize(og, k, w))
    got = sketch(ctx, dg, k, w).to_numpy()
    for a, b in zip(got, exp):
        assert np.array_equal(a, b.astype(a.dtype))


@pytest.mark.parametrize("k,w,fpr", [(24, 1000, 0.025), (24, 100, 0.025), (20, 10, 0.3), (24, 250, 0.9)])
def test_sketch_with_filter(ctx, k, w, fpr):
    "filter-in Bloom filter (indexlr -s): rejected k-mers take the UINT64_MAX sentinel"
    from ntsynt_amd.device import BloomFilter, sketch
    names, seqs = _family(2000 + w, lengths=[60000, 300, 0, 20000], n_frac=0.003)
    rng = np.random.default_rng(w)
    other = []
    for s in seqs:
        a = np.frombuffer(s, dtype=np.uint8).copy()
        hit = rng.random(a.size) < 0.02
        a[hit] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(hit.sum()))]
        # a long stretch unique to this genome: windows there hold only rejected k-mers
        if a.size > 30000:
            a[5000:12000] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=7000)]
        other.append(a.tobytes())
    og = [to_oracle(names, seqs), to_oracle(names, other)]
    dg = [to_device(ctx, names, seqs), to_device(ctx, names, other)]
    nbytes = O.bf_ctor_bytes(O.bf_approx_bytes(og[0].total_bp, fpr))
    obf = O.bf_build(og[1], k, nbytes, prev=O.bf_build(og[0], k, nbytes))
    dbf = BloomFilter(ctx, nbytes, k)
    dbf.insert(dg[0])
    tmp = BloomFilter(ctx, nbytes, k)
    tmp.insert(dg[1])
    dbf.and_(tmp)
    assert np.array_equal(dbf.to_numpy(), obf)
    for o, d in zip(og, dg):
        exp = oracle_flat(O.minimize(o, k, w, obf))
        got = sketch(ctx, d, k, w, dbf).to_numpy()
        assert len(exp[0]) > 0
        for a, b in zip(got, exp):
            assert np.array_equal(a, b.astype(a.dtype))


@pytest.mark.parametrize("k,w,use_bf", [(24, 1000, True), (24, 100, True), (20, 10, False), (129, 50, False)])
def test_sketch_batch_equals_per_genome_and_oracle(ctx, k, w, use_bf):
    """Genome.concat: several resident genomes sketched with one sequence of launches; the list splits into exactly
    the per-genome lists (records never share k-mers), which are the oracle's.  Empty records, records shorter than
    k, N runs and a part without records are in the batch."""
    from ntsynt_amd.device import BloomFilter, Genome, sketch
    parts = [_family(4000 + w, lengths=[60000, 300, 0, 20000], n_frac=0.003),
             _family(4100 + w, lengths=[5, 70001], n_frac=0.0),
             ([], []),
             _family(4200 + w, lengths=[1024, 23, 33000, 0], n_frac=0.01)]
    og = [to_oracle(n, s) for n, s in parts]
    dg = [to_device(ctx, n, s) for n, s in parts]
    obf = dbf = None
    if use_bf:
        nbytes = O.bf_ctor_bytes(O.bf_approx_bytes(og[0].total_bp, 0.3))
        obf = O.bf_build(og[0], k, nbytes)
        for o in og[1:]:
            obf |= O.bf_build(o, k, nbytes)            # (a union, so that every part keeps minimizers)
        obf[::3] = 0                                    # and holes, so that the filter rejects some k-mers everywhere
        dbf = BloomFilter(ctx, nbytes, k)
        dbf.from_numpy(obf)
    batch = Genome.concat(ctx, dg)
    assert batch.total_bp == sum(d.total_bp for d in dg) and len(batch.names) == sum(len(n) for n, _ in parts)
    assert batch.valid_kmers(k) == sum(d.valid_kmers(k) for d in dg)
    got = batch.split_minimizers(*sketch(ctx, batch, k, w, dbf).to_numpy())
    assert len(got) == len(parts)
    total = 0
    for o, d, g in zip(og, dg, got):
        exp = oracle_flat(O.minimize(o, k, w, obf))
        alone = sketch(ctx, d, k, w, dbf).to_numpy()
        total += len(exp[0])
        for a, b, c in zip(g, exp, alone):
            assert np.array_equal(a, b.astype(a.dtype)) and np.array_equal(a, c)
    assert total > 0
    # masks address the batch's record ids
    masks = [(0, 1000, 30000), (int(batch.rec_base[1]) + 1, 100, 50000), (int(batch.rec_base[3]) + 2, 0, 40000)]
    gm = batch.split_minimizers(*sketch(ctx, batch, k, w, dbf, masks).to_numpy())
    for p, d, g in zip(range(len(dg)), dg, gm):
        local = [(r - int(batch.rec_base[p]), s, e) for r, s, e in masks if batch.rec_base[p] <= r < batch.rec_base[p + 1]]
        alone = sketch(ctx, d, k, w, dbf, local).to_numpy()
        for a, c in zip(g, alone):
            assert np.array_equal(a, c)
    batch.free()


@pytest.mark.parametrize("w", [100, 10])
def test_sketch_masked(ctx, w):
    "refinement re-sketch (B5): masks applied on the resident genome == indexlr on the N-masked FASTA"
    from ntsynt_amd.device import sketch
    k = 24
    names, seqs = _family(3000 + w, lengths=[80000, 40000, 100, 0])
    masks = [(0, 1100, 30000), (0, 31000, 31010), (0, 50000, 79990), (1, 0, 500), (1, 20000, 60000), (2, 10, 20),
             (0, 29000, 29500)]
    masked = []
    for i, s in enumerate(seqs):
        b = bytearray(s)
        for r, st, en in masks:
            if r == i:
                en = min(en, len(b))
                b[st:en] = b"N" * max(0, en - st)
        masked.append(bytes(b))
    exp = oracle_flat(O.minimize(to_oracle(names, masked), k, w))
    dg = to_device(ctx, names, seqs)
    got = sketch(ctx, dg, k, w, None, masks).to_numpy()
    for a, b in zip(got, exp):
        assert np.array_equal(a, b.astype(a.dtype))
    # and the unmasked sketch of the same handle is unaffected by the masked call
    exp0 = oracle_flat(O.minimize(to_oracle(names, seqs), k, w))
    got0 = sketch(ctx, dg, k, w).to_numpy()
    for a, b in zip(got0, exp0):
        assert np.array_equal(a, b.astype(a.dtype))


def test_sketch_ties_and_monotone(ctx):
    "homopolymer / low-complexity records: many equal hashes (rightmost-minimum rule) and long runs"
    from ntsynt_amd.device import sketch
    seqs = [b"A" * 5000, b"AC" * 4000, b"ACGT" * 3000 + b"N" + b"TTTT" * 2000, b"G" * 30 + b"N" * 10 + b"C" * 2000]
    names = [f"t{i}" for i in range(len(seqs))]
    og, dg = to_oracle(names, seqs), to_device(ctx, names, seqs)
    for k, w in ((24, 100), (20, 1000), (24, 7)):
        exp = oracle_flat(O.minimize(og, k, w))
        got = sketch(ctx, dg, k, w).to_numpy()
        for a, b in zip(got, exp):
            assert np.array_equal(a, b.astype(a.dtype))


def test_sketch_large_properties(ctx):
    "size-independent properties at 40 Mbp: ordering, gap <= w in valid-k-mer space, idempotence"
    from ntsynt_amd.device import sketch
    rng = np.random.default_rng(77)
    seqs = random_records(rng, [25_000_000, 15_000_000], n_frac=0.0, lower_frac=0.0)
    names = ["a", "b"]
    dg = to_device(ctx, names, seqs)
    k, w = 24, 1000
    h1, rec, pos = sketch(ctx, dg, k, w).to_numpy()
    assert h1.size > 2 * 40_000_000 // (w + 1) * 0.9
    for r in (0, 1):
        p = pos[rec == r].astype(np.int64)
        assert (np.diff(p) > 0).all()
        assert np.diff(p).max() <= w and p[0] < w and (len(seqs[r]) - k - p[-1]) < w
    h1b, recb, posb = sketch(ctx, dg, k, w).to_numpy()
    assert np.array_equal(h1, h1b) and np.array_equal(pos, posb)
    # spot-check against the oracle on a 300 kbp slice boundary-free region: whole record 1 prefix
    sub = seqs[1][:300000]
    exp = O.minimize(O.Genome(["s"], [sub]), k, w)[0]
    m = (rec == 1) & (pos < 300000 - k - w)
    n = int(m.sum())
    assert np.array_equal(pos[m], exp[1][:n]) and np.array_equal(h1[m], exp[0][:n])


@pytest.mark.parametrize("mode,c", [("dense", 0), ("pruned", 64), ("pruned", 4), ("pruned", 1), ("pruned", 100000)])
def test_sketch_modes_identical(ctx, mode, c):
    """dense and pruned sketches are the same function: tiny prune_c forces most windows through the
    uncovered-range fallback, huge prune_c makes every k-mer a candidate"""
    from ntsynt_amd.device import BloomFilter, sketch
    k = 24
    names, seqs = _family(4242, lengths=[150000, 700, 0, 60000, 2500], n_frac=0.004)
    rng = np.random.default_rng(9)
    other = []
    for s in seqs:
        a = np.frombuffer(s, dtype=np.uint8).copy()
        hit = rng.random(a.size) < 0.01
        a[hit] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(hit.sum()))]
        if a.size > 100000:
            a[20000:45000] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=25000)]
        other.append(a.tobytes())
    og = [to_oracle(names, seqs), to_oracle(names, other)]
    dg = [to_device(ctx, names, seqs), to_device(ctx, names, other)]
    nbytes = O.bf_ctor_bytes(O.bf_approx_bytes(og[0].total_bp, 0.025))
    obf = O.bf_build(og[1], k, nbytes, prev=O.bf_build(og[0], k, nbytes))
    dbf = BloomFilter(ctx, nbytes, k)
    dbf.from_numpy(obf)
    ctx.sketch_mode(mode, c)
    try:
        for w in (1000, 300, 64):
            for o, d in zip(og, dg):
                for bf_o, bf_d in ((obf, dbf), (None, None)):
                    exp = oracle_flat(O.minimize(o, k, w, bf_o))
                    got = sketch(ctx, d, k, w, bf_d).to_numpy()
                    for a, b in zip(got, exp):
                        assert np.array_equal(a, b.astype(a.dtype)), (mode, c, w)
                    cand, gaps, gk = ctx.sketch_stats()
                    if mode == "dense":
                        assert (cand, gaps, gk) == (0, 0, 0)
                    elif c == 1 and w == 1000 and bf_o is not None:
                        assert gaps > 0 and gk > 0
        # masked re-sketch through the pruned path as well
        masks = [(0, 5000, 90000), (3, 100, 30000), (0, 120000, 140000)]
        masked = []
        for i, s in enumerate(seqs):
            b = bytearray(s)
            for r, st, en in masks:
                if r == i:
                    b[st:min(en, len(b))] = b"N" * (min(en, len(b)) - st)
            masked.append(bytes(b))
        exp = oracle_flat(O.minimize(to_oracle(names, masked), k, 300, obf))
        got = sketch(ctx, dg[0], k, 300, dbf, masks).to_numpy()
        for a, b in zip(got, exp):
            assert np.array_equal(a, b.astype(a.dtype))
    finally:
        ctx.sketch_mode("auto", 64)


@pytest.mark.parametrize("w,n", [(33, 100_000), (64, 150_000), (300, 800_000)])
def test_uncovered_ranges_device_merge_and_its_fallback(ctx, w, n):
    """Pruned sketch with c = 1: nearly every window is uncovered.  The uncovered ranges' winners are strung together on
    the device (k_gap_collect) when the host expects few of them; with w = 33 and 64 a window tile holds more winners
    than its slot (128), the device flags it and the call is repeated on the general path; w = 300 fits."""
    from ntsynt_amd.device import sketch
    k = 24
    names, seqs = _family(7000 + w, lengths=[n, 0, 40], n_frac=0.0)
    og, dg = to_oracle(names, seqs), to_device(ctx, names, seqs)
    exp = oracle_flat(O.minimize(og, k, w))
    ctx.sketch_mode("pruned", 1)
    try:
        got = sketch(ctx, dg, k, w).to_numpy()
        cand, gaps, gk = ctx.sketch_stats()
        assert gaps > 0 and gk > n // 2
        for a, b in zip(got, exp):
            assert np.array_equal(a, b.astype(a.dtype))
        got2 = sketch(ctx, dg, k, w).to_numpy()          # the context is back on the device-side path afterwards
        for a, b in zip(got2, exp):
            assert np.array_equal(a, b.astype(a.dtype))
    finally:
        ctx.sketch_mode("auto", 64)


@pytest.mark.parametrize("lo,hi", [(40, 400), (30, 110), (1500, 6000)])
def test_fragmented_assembly_sketch_and_bloom(ctx, lo, hi):
    """Thousands of short records: every tile of the select and Bloom kernels spans many runs, lanes have run boundaries
    inside (several per wave; with records of ~70 bases more per tile than the Bloom kernel's side buffer holds).
    Sketch (pruned, forced and automatic) and both Bloom builds against the oracle."""
    from ntsynt_amd.device import BloomFilter, sketch
    rng = np.random.default_rng(lo)
    total = 260_000
    lengths = []
    while sum(lengths) < total:
        lengths.append(int(rng.integers(lo, hi)))
    k = 24
    names, seqs = _family(9000 + lo, lengths=lengths, n_frac=0.002)
    og, dg = to_oracle(names, seqs), to_device(ctx, names, seqs)
    nbytes = 1 << 20
    want_bf = O.bf_build(og, k, nbytes)
    try:
        for mode in ("atomic", "binned"):
            ctx.bf_build_mode(mode)
            bf = BloomFilter(ctx, nbytes, k)
            bf.insert(dg)
            assert np.array_equal(bf.to_numpy(), want_bf), mode
    finally:
        ctx.bf_build_mode("auto")
    keep = want_bf.copy()
    keep[::2] = 0
    bf.from_numpy(keep)
    try:
        for mode, c in (("pruned", 4), ("pruned", 40), ("auto", 0)):
            ctx.sketch_mode(mode, c)
            for w in (300, 64, 20):
                for fo, fd in ((keep, bf), (None, None)):
                    exp = oracle_flat(O.minimize(og, k, w, fo))
                    got = sketch(ctx, dg, k, w, fd).to_numpy()
                    for a, b in zip(got, exp):
                        assert np.array_equal(a, b.astype(a.dtype)), (mode, c, w, fo is None)
    finally:
        ctx.sketch_mode("auto", 64)


@pytest.mark.parametrize("k", [1, 2, 25, 63, 64, 65, 100, 128, 129, 200])
def test_long_and_degenerate_k_end_to_end(ctx, k):
    """k beyond the LDS-staged fast paths (and k = 1): Bloom build (both builds), pruned and dense sketch vs the oracle; odd k and
    the ends of the ranges the partitioned build treats differently (first k-mer from the two-bases LDS table up to 64, from the
    per-base table up to 128, its own kernel beyond)"""
    from ntsynt_amd.device import BloomFilter, sketch
    names, seqs = _family(600 + k, lengths=[90000, 150, 40000, 0, 260], n_frac=0.001)
    og, dg = to_oracle(names, seqs), to_device(ctx, names, seqs)
    nbytes = 1 << 20
    want_bf = O.bf_build(og, k, nbytes)
    try:
        for mode in ("atomic", "binned"):
            ctx.bf_build_mode(mode)
            bf = BloomFilter(ctx, nbytes, k)
            bf.insert(dg)
            assert np.array_equal(bf.to_numpy(), want_bf), mode
            bf.free()
    finally:
        ctx.bf_build_mode("auto")
    dbf = BloomFilter(ctx, nbytes, k)
    dbf.from_numpy(want_bf)
    try:
        for mode in ("dense", "pruned"):
            ctx.sketch_mode(mode, 0)
            for w in (300, 7):
                exp = oracle_flat(O.minimize(og, k, w, want_bf))
                got = sketch(ctx, dg, k, w, dbf).to_numpy()
                for a, b in zip(got, exp):
                    assert np.array_equal(a, b.astype(a.dtype)), (mode, k, w)
    finally:
        ctx.sketch_mode("auto", 0)


@pytest.mark.parametrize("tpw", ["1", "3"])
@pytest.mark.parametrize("k,w,c", [(24, 1000, 16), (20, 700, 12), (32, 400, 6), (24, 250, 9), (31, 1000, 40), (33, 1000, 16)])
def test_select_kernels_agree(ctx_x, monkeypatch, k, w, c, tpw):
    """The two candidate-selection kernels of the pruned sketch (k_hash_select_hi: upper halves rolled, listed k-mers hashed in
    full, probes overlapped with the next tile; k_hash_select: full-width rolling) and the dense path give the oracle's list:
    fragmented records with N runs (tiles that span runs), a satellite repeat (tiles that list more k-mers than a round
    holds), records shorter than k, with and without a filter, one and several tiles per wave (NTS_HI_TPW), both list
    widths (c/w below and above 1/42), k on either side of the kernel's limit (k = 33: full-width kernel in both settings)."""
    ctx = ctx_x            # (environment switches of the experiments build: tests/conftest.py)
    from ntsynt_amd.device import BloomFilter, sketch
    monkeypatch.setenv("NTS_HI_TPW", tpw)
    rng = np.random.default_rng(1234 + k)
    _, clean = _family(177 + k, lengths=[260000, 100000])        # whole tiles inside one run: the pipelined path
    _, ragged = _family(77 + k, lengths=[900, 20, 0, 130000, 41000, 5000, 3000, 2999, 70000], n_frac=0.003)   # an N every ~300 bases
    seqs = list(clean) + list(ragged)
    names = [f"r{i}" for i in range(len(seqs))]
    big = bytearray(seqs[0])
    big[60000:90000] = (b"ACGTTGCA" * 4000)[:30000]          # period-8 satellite: a handful of distinct k-mers, thousands of copies
    big[150000:150300] = b"A" * 300
    seqs[0] = bytes(big)
    other = []
    for s in seqs:
        a = np.frombuffer(s, dtype=np.uint8).copy()
        hit = (rng.random(a.size) < 0.01) & (a != ord("N"))
        a[hit] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(hit.sum()))]
        other.append(a.tobytes())
    og = [to_oracle(names, seqs), to_oracle(names, other)]
    dg = [to_device(ctx, names, seqs), to_device(ctx, names, other)]
    nbytes = O.bf_ctor_bytes(O.bf_approx_bytes(og[0].total_bp, 0.025))
    obf = O.bf_build(og[1], k, nbytes, prev=O.bf_build(og[0], k, nbytes))
    dbf = BloomFilter(ctx, nbytes, k)
    dbf.from_numpy(obf)
    try:
        for o, d in zip(og, dg):
            for bf_o, bf_d in ((obf, dbf), (None, None)):
                exp = oracle_flat(O.minimize(o, k, w, bf_o))
                assert exp[0].size > 100
                for impl in ("hi", "auto", "full"):
                    ctx.sketch_select(impl)
                    ctx.sketch_mode("pruned", c)
                    got = sketch(ctx, d, k, w, bf_d).to_numpy()
                    for a, b in zip(got, exp):
                        assert np.array_equal(a, b.astype(a.dtype)), (impl, k, w, c, bf_o is not None)
                    cand, _, _ = ctx.sketch_stats()
                    assert cand > 0
                    if impl == "hi":
                        # the same without dropping the accepted k-mers that cannot be a window's minimum inside the select
                        # kernel (NTS_SELECT_ELIM=0): same list, more candidates handed to the window kernel
                        monkeypatch.setenv("NTS_SELECT_ELIM", "0")
                        kept = sketch(ctx, d, k, w, bf_d).to_numpy()
                        monkeypatch.delenv("NTS_SELECT_ELIM")
                        for a, b2 in zip(kept, exp):
                            assert np.array_equal(a, b2.astype(a.dtype)), ("no elimination", k, w, c)
                        cand_all, _, _ = ctx.sketch_stats()
                        assert cand <= cand_all           # (tiles inside one run only: few of them in these ragged records)
    finally:
        ctx.sketch_select("auto")
        ctx.sketch_mode("auto", 0)
        dbf.free()
        for d in dg:
            d.free()


@pytest.mark.parametrize("k,w,with_common", [(24, 1000, True), (24, 100, True), (20, 33, False), (32, 400, True)])
def test_sketch_with_filter_out_repeat_filter(ctx, k, w, with_common):
    """indexlr -r: k-mers present in the repeat filter (k-mers seen twice within a genome: nts_bf_insert_repeats, checked
    against the oracle's bits here as well) are rejected next to the ones absent from the common filter; with and without
    the common filter, with hard masks, on records with N runs and a satellite array that fills the repeat filter."""
    from ntsynt_amd.device import BloomFilter, sketch
    rng = np.random.default_rng(k * 1000 + w)
    names, seqs = _family(4000 + w, lengths=[120000, 900, 0, 50000, 2500, 40000], n_frac=0.002)
    seqs = list(seqs)
    big = bytearray(seqs[0])
    unit = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=700))
    big[30000:30000 + 7000] = unit * 10                          # a tandem repeat: its k-mers enter the repeat filter
    big[90000:97000] = big[10000:17000]                          # a dispersed duplicate
    seqs[0] = bytes(big)
    other = []
    for s in seqs:
        a = np.frombuffer(s, dtype=np.uint8).copy()
        hit = (rng.random(a.size) < 0.01) & (a != ord("N"))
        a[hit] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(hit.sum()))]
        other.append(a.tobytes())
    og = [to_oracle(names, seqs), to_oracle(names, other)]
    dg = [to_device(ctx, names, seqs), to_device(ctx, names, other)]
    nbytes = O.bf_ctor_bytes(O.bf_approx_bytes(og[0].total_bp, 0.025))
    obf = O.bf_build(og[1], k, nbytes, prev=O.bf_build(og[0], k, nbytes))
    dbf = BloomFilter(ctx, nbytes, k)
    dbf.from_numpy(obf)
    rep_bytes = 40000
    orep = O.repeat_bf(og, k, rep_bytes)
    drep = BloomFilter(ctx, rep_bytes, k)
    own = BloomFilter(ctx, rep_bytes, k)
    for d in dg:
        own.clear()
        drep.insert_repeats_of(d, own)
    assert np.array_equal(drep.to_numpy(), orep) and O.bf_popcount(orep) > 5000
    try:
        for o, d in zip(og, dg):
            exp = oracle_flat(O.minimize(o, k, w, obf if with_common else None, repeat=orep))
            plain = oracle_flat(O.minimize(o, k, w, obf if with_common else None))
            assert exp[0].size > 50 and not (exp[1].size == plain[1].size and np.array_equal(exp[1], plain[1]))   # the filter bites
            got = sketch(ctx, d, k, w, dbf if with_common else None, repeat=drep).to_numpy()
            for a, b in zip(got, exp):
                assert np.array_equal(a, b.astype(a.dtype))
        masks = [(0, 5000, 60000), (3, 100, 30000)]
        masked = []
        for i, s in enumerate(seqs):
            b = bytearray(s)
            for r, st, en in masks:
                if r == i:
                    b[st:min(en, len(b))] = b"N" * (min(en, len(b)) - st)
            masked.append(bytes(b))
        exp = oracle_flat(O.minimize(to_oracle(names, masked), k, w, obf if with_common else None, repeat=orep))
        got = sketch(ctx, dg[0], k, w, dbf if with_common else None, masks, repeat=drep).to_numpy()
        for a, b in zip(got, exp):
            assert np.array_equal(a, b.astype(a.dtype))
    finally:
        for x in (dbf, drep, own):
            x.free()
        for d in dg:
            d.free()


def test_sketch_pool_equals_one_after_the_other(ctx):
    """device.SketchPool: several genomes sketched at once, each on a context of its own from a thread of its own, give the lists
    sketch() gives one after the other -- with and without hard masks, dense and pruned, more genomes than contexts; and a
    sparse filter, whose summary the contexts share (built once, under the filter's lock)."""
    from ntsynt_amd.device import BloomFilter, SketchPool, sketch
    k, w = 24, 300
    fams = [_family(500 + j, lengths=[300000, 120000, 4000]) for j in range(5)]
    og = [to_oracle(n, s2) for n, s2 in fams]
    dg = [to_device(ctx, n, s2) for n, s2 in fams]
    nbytes = O.bf_ctor_bytes(O.bf_approx_bytes(og[0].total_bp, 0.025))
    obf = O.bf_build(og[0], k, nbytes)
    dbf = BloomFilter(ctx, nbytes, k)
    dbf.from_numpy(obf)
    sparse = obf.copy()
    sparse[np.arange(sparse.size) % 97 != 0] = 0                 # an all-but-empty filter: the summary-first path
    sbf = BloomFilter(ctx, nbytes, k)
    sbf.from_numpy(sparse)
    masks = [[(0, 1000 * (j + 1), 50000 + 1000 * j), (1, 0, 3000)] for j in range(5)]
    pool = SketchPool(ctx, 3)
    try:
        for mode in ("auto", "dense", "pruned"):
            pool.configure(lambda c: c.sketch_mode(mode, 12 if mode == "pruned" else 0))
            for bf_d, mk in ((dbf, None), (dbf, masks), (None, None), (sbf, None)):
                got = pool.sketch(dg, k, w, bf_d, mk)
                for j, mx in enumerate(got):
                    one = sketch(ctx, dg[j], k, w, bf_d, mk[j] if mk else None)
                    for a, b in zip(mx.to_numpy(), one.to_numpy()):
                        assert np.array_equal(a, b), (mode, j)
                    one.free()
                for mx in got:
                    mx.free()
        exp = oracle_flat(O.minimize(og[2], k, w, obf))
        pool.configure(lambda c: c.sketch_mode("auto", 0))
        got = pool.sketch(dg, k, w, dbf)
        for a, b in zip(got[2].to_numpy(), exp):
            assert np.array_equal(a, b.astype(a.dtype))
        for mx in got:
            mx.free()
    finally:
        pool.close()
        ctx.sketch_mode("auto", 0)
        for d in dg:
            d.free()
        dbf.free()
        sbf.free()


@pytest.mark.parametrize("k,w", [(24, 10), (24, 33), (24, 63), (20, 2), (40, 50), (130, 20)])
def test_short_windows_fused_kernel_equals_the_key_array_path_and_the_oracle(ctx, ctx_x, monkeypatch, k, w):
    """w < 64: the window tiles hash and probe their own k-mers (k_window_min<true>, no key array) -- against the two-kernel path
    (k_hash<MODE_KEYS> + k_window_min<false>; NTS_WIN_FUSE=0 in the experiments build) and the oracle, with and without a filter, on
    records with N runs inside tiles (the tile then walks the run table), records shorter than a window, records that end a few
    k-mers into a tile, and k > 128 (the generic walk for every tile)"""
    from ntsynt_amd.device import BloomFilter, sketch
    names, seqs = _family(7000 + 13 * w + k, lengths=[70000, 4096 + w + k - 2, k + w - 2, k + w - 1, 0, 9000, 30000], n_frac=0.004)
    rng = np.random.default_rng(w)
    other = []
    for s_ in seqs:
        a = np.frombuffer(s_, dtype=np.uint8).copy()
        hit = rng.random(a.size) < 0.03
        a[hit] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(hit.sum()))]
        other.append(a.tobytes())
    og = [to_oracle(names, seqs), to_oracle(names, other)]
    nbytes = O.bf_ctor_bytes(O.bf_approx_bytes(og[0].total_bp, 0.05))
    obf = O.bf_build(og[1], k, nbytes, prev=O.bf_build(og[0], k, nbytes))
    for use_filter in (False, True):
        exp = oracle_flat(O.minimize(og[0], k, w, obf if use_filter else None))
        got = {}
        for label, c in (("fused", ctx), ("key array", ctx_x)):
            if label == "key array":
                monkeypatch.setenv("NTS_WIN_FUSE", "0")
            dg = to_device(c, names, seqs)
            bf = None
            if use_filter:
                bf = BloomFilter(c, nbytes, k)
                bf.from_numpy(obf)
            c.sketch_mode("dense")                     # (the every-k-mer path: where short windows go at full size, and where the fusion applies)
            c.profile(1)
            mx = sketch(c, dg, k, w, bf)
            got[label] = mx.to_numpy()
            c.sync()
            hashed, windows = c.timing("hash_probe" if use_filter else "hash_only")[1], c.timing("window_min")[1]
            # the fused pass is timed as the hashing pass and launches no window kernel of its own; the key-array path launches both
            # (a second launch of either: the output segments were sized too small for so short an input, and the pass ran again);
            # k > 128 is not fused (no LDS staging of the bases: k_hash's generic walk + the window kernel)
            # (with a filter sparse enough for its summary to go in front of the probes the key array stays: k_hash_keys_sparse)
            if not use_filter:
                fused_here = label == "fused" and k <= 128
                assert hashed >= 1 and (windows == 0 if fused_here else windows >= 1), (label, hashed, windows)
            c.profile(0)
            c.sketch_mode("auto")
            mx.free()
            if bf is not None:
                bf.free()
            dg.free()
            monkeypatch.delenv("NTS_WIN_FUSE", raising=False)
        assert got["fused"][0].size > 100
        for a, b, e in zip(got["fused"], got["key array"], exp):
            assert np.array_equal(a, b) and np.array_equal(a, e.astype(a.dtype)), (k, w, use_filter)
