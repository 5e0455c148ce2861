"""BASELINE.json's configurations through the HIP path (C1 ... C5; C3 is tests/test_gpu_scale.py).

C1  the reference's own fixtures (tests/expected_result/*.k{20,24}.w1000.tsv as arrays under tests/golden/, the
    KAT sample of its hash:pos:kmer tokens) through nts_hash_all, nts_graph_build and the product engine
    (/root/reference/tests/ntsynt_tests.py:40-59 runs these two parameterisations);
C2  3 x 100 Mbp at 1 %, common filter, full size: pruned == dense == one batch, oracle on slices read back;
C4  one GPU's share of 8 x 3 Gbp at 10 %: one 3 Gbp genome against the AND of eight filters (dense path);
C5  `-d 1.3` parameter set (w_rounds 250 100, indel 50000, merge 100000, block 1000) at 3 x 100 Mbp, FASTA files to TSV,
    byte-identical to the oracle pipeline.
"""
import collections
import os

import numpy as np
import pytest

from oracle import nts_oracle as O
from oracle import synteny_oracle as SO

pytestmark = pytest.mark.gpu

MX_FILES = {
    ("ref", 24): "mx_celegans-chrII-III.fa.k24.w1000.npz",
    ("A", 24): "mx_celegans-chrII-III.A.fa.k24.w1000.npz",
    ("ref", 20): "mx_celegans-chrII-III.fa.k20.w1000.npz",
    ("A", 20): "mx_celegans-chrII-III.A.fa.k20.w1000.npz",
    ("B", 20): "mx_celegans-chrII-III.B.fa.k20.w1000.npz",
}
FASTA = {"ref": "celegans-chrII-III.fa", "A": "celegans-chrII-III.A.fa", "B": "celegans-chrII-III.B.fa"}


@pytest.fixture(scope="module")
def ctx():
    from ntsynt_amd.device import Context
    c = Context(0)
    yield c
    c.close()


# ------------------------------------------------------------------------------------------------ C1
def test_c1_known_answer_kmers_through_nts_hash_all(ctx, golden_dir):
    "9,835 hash:pos:kmer tokens written by the reference's indexlr: each k-mer as a record of its own through the HIP hash"
    from tests.helpers import to_device
    by_k = collections.defaultdict(list)
    with open(os.path.join(golden_dir, "kat_nthash.tsv")) as fh:
        for line in fh:
            k, h1, _pos, kmer, _src = line.rstrip("\n").split("\t")
            by_k[int(k)].append((kmer.encode(), int(h1)))
    assert sum(len(v) for v in by_k.values()) > 9000
    for k, rows in by_k.items():
        seqs = [s for s, _ in rows]
        # half of them lower case: SeqReader folds case (u3), the hash must not change
        seqs = [s.lower() if i % 2 else s for i, s in enumerate(seqs)]
        dev = to_device(ctx, [f"r{i}" for i in range(len(seqs))], seqs)
        h0 = dev.hash_all(k)
        assert h0.size == len(rows)
        got = np.array([O.h1_from_h0(int(h), k) for h in h0.tolist()], dtype=np.uint64)
        assert np.array_equal(got, np.array([h for _, h in rows], dtype=np.uint64))
        dev.free()


def _golden_lists(golden_dir, keys):
    lists, contigs = [], []
    for key in keys:
        z = np.load(os.path.join(golden_dir, MX_FILES[key]))
        contigs.append([str(c) for c in z["contigs"]])
        lists.append((z["h1"].astype(np.uint64), z["contig_idx"].astype(np.uint32), z["pos"].astype(np.uint64)))
    return lists, contigs


@pytest.mark.parametrize("keys,k,stem,counts", [
    ([("ref", 24), ("A", 24)], 24, "celegans-A-ntSynt", (53491, 53523, 53455)),
    ([("ref", 20), ("A", 20), ("B", 20)], 20, "celegans-A-B-ntSynt", (51307, 51372, 51238)),
])
@pytest.mark.parametrize("simplify", [False, True])
def test_c1_reference_minimizers_through_device_graph_and_engine(ctx, golden_dir, tmp_path, keys, k, stem, counts, simplify):
    """The reference's minimizer TSVs -> nts_graph_build -> product engine, initial round: vertex / edge / full-weight
    edge counts of SURVEY.md 8(a) C2, path and block counts (28/29 and 50/54 without bubble removal, 11/15 and 12/16
    with it, the reference default), every block inside one expected (post-refinement, pre-merge) block with the same
    contigs and orientation, and the initial-round TSV equal to the oracle's byte for byte."""
    from ntsynt_amd.graph import build_graph_device, edge_degrees, walk_paths
    from ntsynt_amd.synteny import SyntenyEngine
    lists, contigs = _golden_lists(golden_dir, keys)
    tsvs = [f"{FASTA[key[0]]}.k{k}.w1000.tsv" for key in keys]
    n_paths, n_blocks = {(24, False): (28, 29), (20, False): (50, 54), (24, True): (11, 15), (20, True): (12, 16)}[(k, simplify)]

    def graph_fn(ls, keeps, lids):
        return build_graph_device(ctx, ls, keeps, lids)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        eng = SyntenyEngine(tsvs, contigs, k, 1000, [], 500, 3000, 500, "hip", graph_fn, None, walk_paths,
                            simplify=simplify, degree_fn=edge_degrees)
        ga = graph_fn([lists[i] for i in eng.input_order], None, None)
        assert (ga.v_hash.size, ga.e_u.size, int((ga.e_w == len(keys)).sum())) == counts
        eng._add_graph(ga)
        if simplify:
            eng._simplify(apply_deletions=True)
        eng.e_alive &= eng.e_w >= eng.n
        verts, off = eng._paths()
        assert off.size - 1 == n_paths
        blocks = eng._drop_small(eng._blocks_of_paths((verts, off)), 4)
        assert len(blocks) == n_blocks
        exp = collections.defaultdict(dict)
        for line in open(os.path.join(golden_dir, stem + ".pre-collinear-merge.synteny_blocks.tsv")):
            num, asm, ctg, start, end, ori, _ = line.rstrip("\n").split("\t")
            exp[int(num)][asm] = (ctg, int(start), int(end), ori)
        for b in eng._sorted(blocks):
            mine = {}
            for a in range(eng.G):
                name = SO.MX_SUFFIX.search(eng.files[a]).group(1)
                mine[name] = (eng.contigs[a][b.rec[a]], eng._start(b, a), eng._end(b, a), b.ori[a])
            hits = [num for num, e in exp.items()
                    if all(e[a][0] == mine[a][0] and e[a][1] <= mine[a][1] and mine[a][2] <= e[a][2] and e[a][3] == mine[a][3]
                           for a in mine)]
            assert len(hits) == 1, mine
        # whole initial round once more through run() against the oracle on the same lists
        eng2 = SyntenyEngine(tsvs, contigs, k, 1000, [], 500, 3000, 500, "hip", graph_fn, None, walk_paths,
                             simplify=simplify, degree_fn=edge_degrees)
        got = eng2.run(lists)["hip.synteny_blocks.tsv"]
        tables = {}
        for tsv, (h1, rec, pos), names in zip(tsvs, lists, contigs):
            recs = [(c, []) for c in names]
            for ci, h, p in zip(rec.tolist(), h1.tolist(), pos.tolist()):
                recs[ci][1].append((str(h), p))
            tables[tsv] = SO.mx_tables_from_tokens(recs)
        ora = SO.SyntenyOracle(list(tables), {}, k, 1000, [], 500, 3000, 500, "ora", simplify=simplify)
        ora.load(tables)
        want = ora.main()["ora.synteny_blocks.tsv"]
        assert got == want and len(got.splitlines()) == n_blocks * len(keys)
    finally:
        os.chdir(cwd)


# ------------------------------------------------------------------------------------------------ C2
def test_c2_three_100mbp_genomes_with_filter_full_size(ctx):
    from ntsynt_amd.device import BloomFilter, Genome, bf_size_bytes, sketch
    k, w, contigs, total = 24, 1000, 4, 100_000_000
    genomes = [Genome.synth(ctx, total, contigs, 20240207, 50 + j, 0.005) for j in range(3)]
    approx, nbytes = bf_size_bytes(total, 0.025)
    assert approx == 493723627
    common = BloomFilter(ctx, nbytes, k)
    common.insert(genomes[0])
    tmp = BloomFilter(ctx, nbytes, k)
    for g in genomes[1:]:
        tmp.clear()
        tmp.insert(g)
        common.and_(tmp)
    tmp.free()
    bits = common.to_numpy()
    res = {}
    for mode in ("pruned", "dense"):
        ctx.sketch_mode(mode)
        res[mode] = [sketch(ctx, g, k, w, common).to_numpy() for g in genomes]
    ctx.sketch_mode("auto")
    for a, b in zip(res["pruned"], res["dense"]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    # the three assemblies as one batch genome (what pipeline.py does below 1 Gbp): the same three lists
    batch = Genome.concat(ctx, genomes)
    parts = batch.split_minimizers(*sketch(ctx, batch, k, w, common).to_numpy())
    for a, b in zip(parts, res["dense"]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    batch.free()
    n_slice = 1_200_000
    for gi, g in enumerate(genomes):
        h1, rec, pos = res["pruned"][gi]
        assert 0.9 * 2 * total / (w + 1) < h1.size < 1.4 * 2 * total / (w + 1)
        for r in (0, contigs - 1):
            seq = g.download(int(g.rec_off[r]), n_slice).tobytes()
            exp = O.minimize(O.Genome(["s"], [seq]), k, w, bits)[0]
            m = (rec == r) & (pos < n_slice - k - w)
            n = int(m.sum())
            assert n > 1000
            assert np.array_equal(pos[m], exp[1][:n]) and np.array_equal(h1[m], exp[0][:n])
    # the filter against the oracle's cascade on a slice family: bits of shared k-mers are set, occupancy as sized
    assert 0.015 < O.bf_fpr(bits) < 0.025
    for g in genomes:
        g.free()
    common.free()


# ------------------------------------------------------------------------------------------------ C4
def test_c4_one_gpu_share_of_eight_divergent_3gbp_genomes(ctx):
    """Config 4 as one rank sees it: its own 3 Gbp genome, the common filter = AND of the eight genomes' filters (10 %
    pairwise divergence: a 24-mer survives in all eight with probability 0.95^192, so the filter is all but empty and
    the library looks every k-mer up in the filter's L2-resident summary first and takes the accepted k-mers as the candidate
    list: k_hash_accept*; the every-k-mer-probed kernels and the forced pruned path are compared with it below).  Properties +
    oracle on slices read back from HBM."""
    from ntsynt_amd.device import BloomFilter, Genome, bf_size_bytes, sketch
    k, w, contigs, total = 24, 1000, 24, 3_000_000_000
    _, nbytes = bf_size_bytes(total, 0.025)
    common = BloomFilter(ctx, nbytes, k)
    tmp = BloomFilter(ctx, nbytes, k)
    mine = None
    for j in range(8):
        g = Genome.synth(ctx, total, contigs, 20240207, 400 + j, 0.05)
        if j == 0:
            common.insert(g)
            occ0 = common.get_fpr()
            mine = g
        else:
            tmp.clear()
            tmp.insert(g)
            common.and_(tmp)
            g.free()
    tmp.free()
    occ = common.get_fpr()
    assert abs(occ0 - 0.025) < 0.001 and occ < 1e-4          # shared k-mers 5e-5 of 0.025, chance bits 0.025^8
    out = {}
    for mode in ("auto", "dense", "pruned"):
        ctx.sketch_mode(mode)
        out[mode] = sketch(ctx, mine, k, w, common).to_numpy()
        if mode == "auto":
            assert ctx.sketch_summary() >= 7            # the all-but-empty filter is probed through its L2-resident summary
    ctx.sketch_summary("no-lds")
    out["no-lds"] = sketch(ctx, mine, k, w, common).to_numpy()     # summary in the L2, no folded copy in LDS
    ctx.sketch_summary("never")
    ctx.sketch_mode("dense")
    out["plain"] = sketch(ctx, mine, k, w, common).to_numpy()      # every k-mer probed in HBM
    assert ctx.sketch_summary() == 0
    ctx.sketch_summary("auto")
    ctx.sketch_mode("auto")
    for mode in ("dense", "pruned", "plain", "no-lds"):
        for x, y in zip(out["auto"], out[mode]):
            assert np.array_equal(x, y)
    h1, rec, pos = out["auto"]
    assert 1000 < h1.size < 1_000_000                          # ~ 3e9 * 5e-5 accepted k-mers, each a minimizer of its windows
    bits = common.to_numpy()
    n_slice = 3_000_000
    seen = 0
    for r in (0, 11, contigs - 1):
        seq = mine.download(int(mine.rec_off[r]), n_slice).tobytes()
        exp = O.minimize(O.Genome(["s"], [seq]), k, w, bits)[0]
        m = (rec == r) & (pos < n_slice - k - w)
        n = int(m.sum())
        seen += n
        assert np.array_equal(pos[m], exp[1][:n]) and np.array_equal(h1[m], exp[0][:n])
    assert seen > 50
    mine.free()
    common.free()


def test_sparse_filter_summary_path_matches_oracle(ctx_x, monkeypatch):
    "the summary-first dense pass (csrc: k_hash_keys_sparse) on ragged records with N runs, against the plain pass and the oracle"
    ctx = ctx_x            # (environment switches of the experiments build: tests/conftest.py)
    from ntsynt_amd import synth
    from ntsynt_amd.device import BloomFilter, bf_size_bytes, sketch
    from tests.helpers import oracle_flat, to_device
    k = 24
    anc = synth.make_ancestor(1_500_000, 5, seed=71)
    anc += [anc[0][:30], anc[1][:23], anc[2][:1023]]                    # records around k and around a window
    fam = [synth.derive_genome(anc, 0.10, j, seed=71, structural=False, n_runs=True) for j in range(4)]
    dev, ora = [], []
    for contigs in fam:
        seqs = [c.tobytes() for c in contigs]
        names = [f"c{i}" for i in range(len(seqs))]
        dev.append(to_device(ctx, names, seqs))
        ora.append(O.Genome(names, seqs))
    _, nbytes = bf_size_bytes(dev[0].total_bp, 0.025)
    common = BloomFilter(ctx, nbytes, k)
    common.insert(dev[0])
    tmp = BloomFilter(ctx, nbytes, k)
    for g in dev[1:]:
        tmp.clear()
        tmp.insert(g)
        common.and_(tmp)
    tmp.free()
    bits = common.to_numpy()
    assert 0 < O.bf_fpr(bits) < 0.002
    for w in (40, 300, 1000):
        for d, o in zip(dev[:2], ora[:2]):
            exp = oracle_flat(O.minimize(o, k, w, bits))
            ctx.sketch_summary("never")
            ctx.sketch_mode("dense")
            b = sketch(ctx, d, k, w, common).to_numpy()                # every k-mer probed in HBM
            # "dense": keys + window kernel behind the summary; "auto": the accepted k-mers as the candidate list, with the folded
            # copy of the filter in LDS first (k_hash_accept4) or without it (k_hash_accept)
            # copy of the filter in LDS first (k_hash_accept4r: bases in registers; k_hash_accept4: bases staged in LDS) or without it
            for smode, mode, reg in (("auto", "dense", "1"), ("auto", "auto", "1"), ("auto", "auto", "0"), ("no-lds", "auto", "1")):
                monkeypatch.setenv("NTS_ACCEPT_REG", reg)
                ctx.sketch_summary(smode)
                ctx.sketch_mode(mode)
                a = sketch(ctx, d, k, w, common).to_numpy()
                assert ctx.sketch_summary() >= 7
                for x, y in zip(a, b):
                    assert np.array_equal(x, y)
                assert np.array_equal(a[0], exp[0]) and np.array_equal(a[2], exp[2]) and a[0].size > 0
    ctx.sketch_summary("auto")
    ctx.sketch_mode("auto")
    # masked re-sketch (refinement rounds) through the same path
    ctx.sketch_summary("auto")
    masks = [(0, 1000, 200_000), (3, 0, 50_000)]
    got = sketch(ctx, dev[0], k, 40, common, masks).to_numpy()
    seqs = []
    for r in range(len(ora[0].names)):
        buf = bytearray(ora[0].record(r))
        for mr, s0, e0 in masks:
            if mr == r:
                buf[s0:e0] = b"N" * (min(e0, len(buf)) - s0)
        seqs.append(bytes(buf))
    exp = oracle_flat(O.minimize(O.Genome(ora[0].names, seqs), k, 40, bits))
    assert np.array_equal(got[0], exp[0]) and np.array_equal(got[2], exp[2])
    ctx.sketch_mode("auto")
    for g in dev:
        g.free()
    common.free()


# ------------------------------------------------------------------------------------------------ C5
@pytest.mark.parametrize("mbp,per_genome", [(100, False), (40, True)])
def test_c5_parameter_set_at_100mbp_files_to_tsv(tmp_path, monkeypatch, mbp, per_genome):
    """`ntSynt -d 1.3` resolves to w_rounds 250 100, indel 50000, merge 100000, block 1000 (bin/ntSynt:92-94): that
    parameter set on 3 x 100 Mbp FASTA files, end to end, against the oracle pipeline.  per_genome: the batch sketch switched
    off (GpuBackend.BATCH_BELOW_BP = 0), i.e. one launch sequence per genome and refinement round and the lists taken apart
    nowhere -- the path every assembly of 1 Gbp and more takes (at full size: scripts/e2e_oracle_check.py,
    profiles/r03_e2e_oracle.json)."""
    from ntsynt_amd import cli, pipeline, synth
    monkeypatch.setattr(pipeline.GpuBackend, "BATCH_BELOW_BP", 0 if per_genome else 1 << 30)
    if per_genome:
        monkeypatch.setenv("NTS_SKETCH_POOL", "3")          # ... and the three genomes of a round sketched at once (device.SketchPool)
    paths = synth.make_family(str(tmp_path), 3, mbp * 1_000_000, 6, 0.013, seed=77, micro=12)
    parser = cli.build_parser()
    a = parser.parse_args(paths + ["-d", "1.3", "-p", "c5"])
    cli.resolve(parser, a)
    assert (a.w_rounds, a.indel, a.merge, a.block_size) == ([250, 100], 50000, 100000, 1000)
    kw = dict(k=a.k, w=a.w, fpr=a.fpr, prefix="c5", w_rounds=a.w_rounds, indel=a.indel, merge=a.merge, block_size=a.block_size)
    cwd = os.getcwd()
    try:
        os.makedirs(tmp_path / "hip")
        os.makedirs(tmp_path / "ora")
        os.chdir(tmp_path / "hip")
        eng = pipeline.run(paths, log=lambda *x: None, write_mx_tsv=False, **kw)
        os.chdir(tmp_path / "ora")
        ora = SO.run_pipeline(paths, threads=os.cpu_count(), write_mx_tsv=False, **kw)
    finally:
        os.chdir(cwd)
    for name in ("c5.synteny_blocks.tsv", "c5.pre-collinear-merge.synteny_blocks.tsv"):
        assert eng.outputs[name] == ora.outputs[name], name
    assert len(eng.outputs["c5.synteny_blocks.tsv"].splitlines()) >= 3 * 6


# ------------------------------------------------------------------------------------------------ exchanges, one rank
def test_comm_layer_on_one_rank(ctx):
    """What a 1-GPU box can show of the RCCL layer: librccl loads, a communicator of one rank comes up, the two exchange
    calls are identities there (AND all-reduce in place, all-gather = copies of the rank's own lists)."""
    from ntsynt_amd.device import BloomFilter, Comm, Genome, Minimizers, sketch
    comm = Comm(ctx, 1, 0, lambda ident: ident)
    g = Genome.synth(ctx, 4_000_000, 2, 7, 8, 0.0)
    bf = BloomFilter(ctx, 1 << 20, 24, world=1)
    bf.insert(g)
    before = bf.to_numpy()
    comm.allreduce_and(bf)
    assert np.array_equal(bf.to_numpy(), before)
    mx = sketch(ctx, g, 24, 100, bf)
    ref = mx.to_numpy()
    other = Minimizers.from_numpy(ctx, ref[0][:10], ref[1][:10], ref[2][:10])
    got = comm.allgather_minimizers([mx, other], [1, 0], 2)
    for x, y in zip(got[1].to_numpy(), ref):
        assert np.array_equal(x, y)
    for x, y in zip(got[0].to_numpy(), ref):
        assert np.array_equal(x, y[:10])
    # a sharded allocation (world 8 layout) is the same filter
    sh = BloomFilter(ctx, 1 << 20, 24, world=8)
    sh.insert(g)
    assert np.array_equal(sh.to_numpy(), before)
    ones = BloomFilter(ctx, 1000, 24, world=3, ones=True)
    assert ones.popcount() == 8000
    for h in (mx, other, *got):
        h.free()
    for h in (bf, sh, ones):
        h.free()
    g.free()
    comm.close()


# ------------------------------------------------------------------------------------------------ unpinned btllib details
@pytest.mark.parametrize("rounding", ["down", "none"])
def test_bloom_rounding_switch_matches_oracle(ctx, tmp_path, rounding):
    """SURVEY.md 8(c) u1 as a switch: with the constructor rounding 'down' or 'none' the modulus of every bit index
    changes; HIP and oracle must agree on the filter bits, the sketch and the whole pipeline for each setting."""
    from ntsynt_amd import pipeline, synth
    from ntsynt_amd.device import BloomFilter, bf_size_bytes
    from tests.helpers import oracle_flat, to_device
    from ntsynt_amd.device import sketch
    paths = synth.make_family(str(tmp_path), 2, 1_200_003, 2, 0.01, seed=91, micro=4)
    genomes = [O.read_fasta(p) for p in paths]
    approx, nbytes = bf_size_bytes(genomes[0].total_bp, 0.025, rounding)
    assert nbytes == O.bf_ctor_bytes(approx, rounding)
    assert nbytes != bf_size_bytes(genomes[0].total_bp, 0.025)[1]
    if rounding == "none":
        assert nbytes % 8 != 0                                   # the case the default rounding never produces
    for mode in ("atomic", "binned"):
        ctx.bf_build_mode(mode)
        dev = [to_device(ctx, g.names, [g.record(i) for i in range(len(g.names))]) for g in genomes]
        bf = BloomFilter(ctx, nbytes, 24)
        bf.insert(dev[0])
        other = BloomFilter(ctx, nbytes, 24)
        other.insert(dev[1])
        bf.and_(other)
        want = O.bf_build(genomes[1], 24, nbytes, prev=O.bf_build(genomes[0], 24, nbytes))
        assert np.array_equal(bf.to_numpy(), want), mode
        for d, g in zip(dev, genomes):
            for sk_mode in ("pruned", "dense"):
                ctx.sketch_mode(sk_mode)
                got = sketch(ctx, d, 24, 300, bf).to_numpy()
                exp = oracle_flat(O.minimize(g, 24, 300, want))
                assert np.array_equal(got[0], exp[0]) and np.array_equal(got[2], exp[2])
        ctx.sketch_mode("auto")
        other.free()
        bf.free()
        for d in dev:
            d.free()
    ctx.bf_build_mode("auto")
    kw = dict(k=24, w=300, w_rounds=[100, 10], indel=500, merge=3000, block_size=300, bf_rounding=rounding)
    cwd = os.getcwd()
    try:
        os.makedirs(tmp_path / "hip")
        os.makedirs(tmp_path / "ora")
        os.chdir(tmp_path / "hip")
        eng = pipeline.run(paths, prefix="r", log=lambda *x: None, ctx=ctx, **kw)
        os.chdir(tmp_path / "ora")
        ora = SO.run_pipeline(paths, prefix="r", **kw)
    finally:
        os.chdir(cwd)
    assert eng.outputs["r.synteny_blocks.tsv"] == ora.outputs["r.synteny_blocks.tsv"]
    bits, _ = pipeline.read_bf(str(tmp_path / "hip" / "r.common.bf"))
    assert np.array_equal(bits, ora.bf) and bits.size == nbytes


# ------------------------------------------------------------------------------------------------ F4: experimental repeat filter
def test_repeat_filter_matches_oracle(ctx, tmp_path):
    """bin/ntsynt_make_repeat_bfs.py:53-69: filter of the k-mers (bits) hit at least twice within a genome, over two genomes,
    device vs the oracle's sequential restatement; and the command-line tool writes that filter."""
    import subprocess
    import sys
    from ntsynt_amd import synth
    from ntsynt_amd.device import BloomFilter
    from ntsynt_amd.pipeline import read_bf
    from tests.helpers import to_device
    paths = synth.make_family(str(tmp_path), 2, 600_000, 3, 0.02, seed=31, micro=25, n_runs=True, soft_mask=True)   # micro: copies => repeats
    genomes = [O.read_fasta(p) for p in paths]
    k, nbytes = 24, 1 << 20
    want = O.repeat_bf(genomes, k, nbytes)
    assert 100 < O.bf_popcount(want) < nbytes * 4
    rep = BloomFilter(ctx, nbytes, k)
    own = BloomFilter(ctx, nbytes, k)
    for g in genomes:
        d = to_device(ctx, g.names, [g.record(i) for i in range(len(g.names))])
        own.clear()
        rep.insert_repeats_of(d, own)
        d.free()
    assert np.array_equal(rep.to_numpy(), want)
    rep.free()
    own.free()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, os.path.join(root, "bin", "ntsynt_make_repeat_bfs"), "--genome", *paths, "-k", str(k), "--bf", "1048576B",
                    "-p", str(tmp_path / "rep")], check=True, env=dict(os.environ, PYTHONPATH=root), stdout=subprocess.DEVNULL)
    bits, kk = read_bf(str(tmp_path / "rep.bf"))
    assert kk == k and np.array_equal(bits, want)


@pytest.mark.gpu
@pytest.mark.parametrize("k,nbytes", [(24, 1 << 18), (20, 100_003 * 8), (40, 1 << 16), (70, 1 << 16)])
def test_minimizer_list_screened_against_a_repeat_filter(ctx, k, nbytes):
    """nts_mx_screen (stage 3's `--filter Filter`: ntJoin's read_minimizers(file, repeat_bf)): the minimizers whose k-mer the filter holds
    leave the list, the others stay in order -- against the oracle's filter test on the k-mer text; soft-masked and ragged records"""
    from ntsynt_amd.device import BloomFilter, sketch
    from tests.helpers import random_records, to_device
    rng = np.random.default_rng(k)
    seqs = random_records(rng, [50_000, 0, k - 1, k, 3000, 120_000, 700], n_frac=0.01)
    names = [f"r{i}" for i in range(len(seqs))]
    g = to_device(ctx, names, seqs)
    og = O.Genome(names, seqs)
    want_bf = O.bf_build(og, k, nbytes)                    # (every k-mer of the genome: all minimizers would go)
    bits = np.unpackbits(want_bf, bitorder="little")
    bits[rng.random(bits.size) < 0.5] = 0                  # half of it: some minimizers stay, some go
    half = np.packbits(bits, bitorder="little")
    rep = BloomFilter(ctx, nbytes, k)
    rep.from_numpy(half)
    mx = sketch(ctx, g, k, 50)
    h1, rec, pos = mx.to_numpy()
    keep = np.array([not O.bf_contains(half, O.hash_kmer(seqs[r][p:p + k])[0]) for r, p in zip(rec.tolist(), pos.tolist())], dtype=bool)
    assert 0.2 < keep.mean() < 0.8 and keep.size > 1000
    got = mx.screened(g, k, rep)
    a, b, c = got.to_numpy()
    assert np.array_equal(a, h1[keep]) and np.array_equal(b, rec[keep]) and np.array_equal(c, pos[keep])
    empty = mx.screened(g, k, BloomFilter(ctx, nbytes, k, ones=True))
    assert len(empty) == 0
    for m in (mx, got, empty):
        m.free()
    rep.free()
    g.free()


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [143_467_638 * 8, (1 << 32) + 1, 14_811_708_827 * 8, (1 << 37) - 1, (1 << 38) - 3, (1 << 38) + 5,
                                  (1 << 40) + 12345, 2, 3, (1 << 32), (1 << 32) - 1])
def test_filter_index_arithmetic_all_forms(ctx, bits):
    """h mod bits as the device computes it (nts_mod_indices -> nts::FastMod, every form the filter size admits) against Python
    integers: the generic 64 x 64 form, the short form for bits > 2^32 and the shorter one below 2^38 that every human-scale
    filter takes (btllib: hashes[0] % array_bits)."""
    import ctypes
    rng = np.random.default_rng(bits % 1000003)
    h = rng.integers(0, 1 << 64, size=200_000, dtype=np.uint64)
    edge = [0, 1, bits - 1, bits, bits + 1, 2 * bits - 1, 2 * bits, (1 << 64) - 1, (1 << 64) - bits, ((1 << 64) // bits) * bits,
            ((1 << 64) // bits) * bits - 1, (1 << 63), (1 << 32) - 1, (1 << 32)]
    h[:len(edge)] = np.array([e % (1 << 64) for e in edge], dtype=np.uint64)
    want = np.array([int(x) % bits for x in h], dtype=np.uint64)
    for form in (-1, 0, 1, 2):
        out = np.empty_like(h)
        ctx.check(ctx.lib.nts_mod_indices(ctx.h, bits, form, h.ctypes.data_as(ctypes.c_void_p), h.size,
                                          out.ctypes.data_as(ctypes.c_void_p)), "nts_mod_indices")
        assert np.array_equal(out, want), (bits, form)
