"""The bench family's generator (nts_genome_synth_plan: structural events as a tiling of ancestor pieces, generated in HBM)
against a numpy evaluation of the same plan, and the product pipeline on such a family against the oracle pipeline -- the
block rules the substitution-only family never reached (indel cuts, orientation changes, contig changes, merges)."""
import os

import numpy as np
import pytest

from ntsynt_amd import synth
from oracle import nts_oracle as O
from oracle import synteny_oracle as SO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from ntsynt_amd.device import Context
    c = Context(0)
    yield c
    c.close()


CODES = np.frombuffer(b"ACGTN", dtype=np.uint8)


@pytest.mark.parametrize("j,n_runs", [(0, False), (1, False), (2, True)])
def test_device_family_equals_its_plan(ctx, j, n_runs):
    from ntsynt_amd.device import BloomFilter, Genome, bf_size_bytes, sketch
    plan = synth.structural_plan(3, 400_000, j, seed=99, n_runs=n_runs, indel_bp=(50, 3000), micro=10, micro_bp=(300, 3000), micro_shift=20000,
                                 small_indels=20)
    rec_len, pieces = plan
    assert pieces["flags"].max() > 0                                   # the plan is not the identity
    g = Genome.synth_plan(ctx, plan, 99, 1000 + j, 0.004)
    assert g.total_bp == int(rec_len.sum())
    got = g.download(0, g.total_bp)
    exp = CODES[synth.plan_bases(plan, 99, 1000 + j, 0.004)]
    assert np.array_equal(got, exp)
    # the table of valid stretches the library derives from the pieces: the sketch agrees with the oracle on the same text
    seqs = [exp[int(o):int(o) + int(n)].tobytes() for o, n in zip(g.rec_off, g.rec_len)]
    og = O.Genome(list(g.names), seqs)
    k, w = 24, 200
    _, nbytes = bf_size_bytes(g.total_bp, 0.025)
    bf = BloomFilter(ctx, nbytes, k)
    bf.insert(g)
    obf = O.bf_build(og, k, nbytes)
    assert np.array_equal(bf.to_numpy(), obf)
    h1, rec, pos = sketch(ctx, g, k, w, bf).to_numpy()
    ref = O.minimize(og, k, w, obf)
    assert np.array_equal(h1, np.concatenate([r[0] for r in ref])) and np.array_equal(pos, np.concatenate([r[1] for r in ref]))
    g.free()


def test_rejects_a_plan_that_does_not_tile(ctx):
    from ntsynt_amd.device import Genome, NtsError
    rec_len, pieces = synth.structural_plan(2, 100_000, 1, seed=5)
    bad = pieces.copy()
    bad["dst"][1] += 1
    with pytest.raises(NtsError):
        Genome.synth_plan(ctx, (rec_len, bad), 5, 6, 0.01)


@pytest.mark.parametrize("erode_on_host", [False, True])
def test_pipeline_on_a_structural_family_matches_the_oracle(ctx_x, tmp_path, monkeypatch, erode_on_host):
    """three genomes with inversions, translocations, indels and soft-masked stretches, written to FASTA from HBM like
    bench.py's e2e leg does; both synteny TSVs byte-identical to the oracle pipeline's, and the rules have fired"""
    ctx = ctx_x            # (environment switches of the experiments build: tests/conftest.py)
    import bench
    from ntsynt_amd import pipeline
    from ntsynt_amd.device import Genome
    if erode_on_host:       # every erosion walk through nts_engine_erode's host path (the one for walks over branching vertices)
        monkeypatch.setenv("NTS_ERODE_HOST", "1")
    paths = []
    for j in range(3):
        plan = synth.structural_plan(4, 1_500_000, j, seed=31, inversions=3, translocations=1, indels=10, indel_bp=(100, 3000),
                                     micro=25, micro_bp=(1500, 6000), micro_shift=30000, small_indels=60)
        g = Genome.synth_plan(ctx, plan, 31, 1000 + j, 0.005)
        p = str(tmp_path / f"syn{j}.fa")
        bench.write_fasta_from_device(g, p, soft_mask_seed=4000 + j)
        g.free()
        paths.append(p)
    kw = dict(k=24, w=200, w_rounds=[100, 20], indel=400, merge=2000, block_size=200, prefix="s")
    cwd = os.getcwd()
    try:
        os.makedirs(tmp_path / "hip")
        os.makedirs(tmp_path / "ora")
        os.chdir(tmp_path / "hip")
        eng = pipeline.run(paths, log=lambda *a: None, ctx=ctx, **kw)
        os.chdir(tmp_path / "ora")
        ora = SO.run_pipeline(paths, threads=4, **kw)
    finally:
        os.chdir(cwd)
    for name in ("s.synteny_blocks.tsv", "s.pre-collinear-merge.synteny_blocks.tsv"):
        assert eng.outputs[name] == ora.outputs[name], name
    text = eng.outputs["s.synteny_blocks.tsv"]
    assert "-" in {ln.split("\t")[5] for ln in text.splitlines()}, "no reversed block: the inversions left no trace"
    print(eng.stats)
    assert eng.stats["indel_cuts"] > 0 and eng.stats["merged"] > 0 and eng.stats["eroded_edges"] > 0 and eng.stats["bubbles"] > 0, eng.stats
    assert len(text.splitlines()) // 3 >= 10


# ---- the assembly-like family (BASELINE config 5 stands on mammalian assemblies, which are not in the container) -----------------
def test_assembly_like_family_equals_its_plan(ctx):
    """nts_genome_synth_plan_ex (interspersed repeat families as a function of the ancestor coordinate, satellite arrays, segmental
    duplications, scaffolds with a tail of short ones, N gaps) against synth.plan_bases, the numpy statement of the same generator"""
    from ntsynt_amd.device import Genome
    for j in range(3):
        plan = synth.realistic_plan(3, 700_000, j, seed=17, n_scaffolds=12 + 9 * j, n_tail=15, n_gaps=20)
        rec_len, pieces, names = plan
        assert (pieces["flags"] & synth.TANDEM).any() and (pieces["flags"] & synth.NRUN).any() and (pieces["flags"] & synth.REVCOMP).any()
        g = Genome.synth_plan(ctx, plan, 17, 1000 + j, 0.0065, rep=synth.REPEATS, names=names)
        assert g.total_bp == int(rec_len.sum()) and len(g.names) == rec_len.size
        got = g.download(0, g.total_bp)
        exp = CODES[synth.plan_bases(plan, 17, 1000 + j, 0.0065, rep=synth.REPEATS)]
        assert np.array_equal(got, exp), j
        g.free()
    # the ancestor really is repetitive: a 24-mer drawn from a young SINE copy occurs many times
    anc = synth.plan_bases((np.array([400_000], dtype=np.uint64), synth._pieces_array([[0, 400_000, 0, 0]])), 17, 1, 0.0, rep=synth.REPEATS)
    kmers = np.lib.stride_tricks.sliding_window_view(anc, 24)[::7]
    packed = (kmers.astype(np.uint64) << (2 * np.arange(24, dtype=np.uint64))).sum(axis=1)
    _, counts = np.unique(packed, return_counts=True)
    assert counts.max() >= 3, counts.max()


_ORACLE_RUNS = {}


def _assembly_like_files(ctx, tmp_path, n_chrom, chrom_bp, seed, scaffolds):
    import bench
    from ntsynt_amd.device import Genome
    paths, dup = [], []
    for j in range(3):
        plan = synth.realistic_plan(n_chrom, chrom_bp, j, seed=seed, n_scaffolds=scaffolds[j], n_tail=scaffolds[j], n_gaps=2 * scaffolds[j], sat_scale=0.4, indel_bp=(1000, 90000))
        g = Genome.synth_plan(ctx, plan, seed, 1000 + j, 0.0065, rep=synth.REPEATS, names=plan[2])        # 1.3 % pairwise
        p = str(tmp_path / f"asm{j}.fa")
        bench.write_fasta_from_device(g, p, soft_mask_seed=4000 + j, half_lower=True, line_width=(0, 80, 61)[j])
        g.free()
        paths.append(p)
    return paths


@pytest.mark.parametrize("batch", [True, False])
def test_pipeline_on_an_assembly_like_family_matches_the_oracle(ctx, tmp_path, monkeypatch, batch):
    """BASELINE config 5's parameter set (ntSynt -d 1.3: w 1000, --w_rounds 250 100, indel 50000, merge 100000, block 1000;
    /root/reference bin/ntSynt:92-94, README.md:157) on three assembly-like genomes of ~56 Mbp: FASTA files (half the bases lower
    case, one single-line, two wrapped) -> minimizer TSVs and both synteny TSVs byte-identical to the oracle pipeline's, through the
    one-batch sketch and through one launch sequence per genome; and the run went where an i.i.d. family never goes: minimizers
    that occur twice within an assembly (row C1's drop), Bloom buckets that overflow, select tiles that list more than their
    slots hold."""
    from ntsynt_amd import pipeline
    monkeypatch.setenv("NTS_BATCH_BELOW_BP", str(1 << 30) if batch else "0")
    monkeypatch.setattr(pipeline.GpuBackend, "BATCH_BELOW_BP", (1 << 30) if batch else 0)
    paths = _assembly_like_files(ctx, tmp_path, 6, 9_000_000, 23, (40, 150, 400))
    kw = dict(k=24, w=1000, w_rounds=[250, 100], indel=50000, merge=100000, block_size=1000, prefix="c5")
    seen = {"direct": 0, "many": 0}
    real_insert_and, real_sketch_dev = pipeline.GpuBackend.bf_insert_and, pipeline.GpuBackend.sketch_dev

    def spy_insert_and(self, acc, genome):
        real_insert_and(self, acc, genome)
        seen["direct"] += self.ctx.path_stats()["bf_direct_indices"]

    def spy_sketch_dev(self, *a, **k_):
        out = real_sketch_dev(self, *a, **k_)
        seen["many"] += self.ctx.path_stats()["sketch_many_listed"]
        return out
    monkeypatch.setattr(pipeline.GpuBackend, "bf_insert_and", spy_insert_and)
    monkeypatch.setattr(pipeline.GpuBackend, "sketch_dev", spy_sketch_dev)
    cwd = os.getcwd()
    try:
        os.makedirs(tmp_path / "hip")
        os.makedirs(tmp_path / "ora")
        os.chdir(tmp_path / "hip")
        eng = pipeline.run(paths, log=lambda *a: None, ctx=ctx, **kw)
        hip_files = {n: open(n, "rb").read() for n in os.listdir(".") if n.endswith(".tsv") or n.endswith(".fai")}
        os.chdir(tmp_path / "ora")
        if "c5" not in _ORACLE_RUNS:              # (the family is deterministic: the oracle runs once for both parametrisations)
            ora = SO.run_pipeline(paths, threads=8, **kw)
            _ORACLE_RUNS["c5"] = (dict(ora.outputs), {n: open(n, "rb").read() for n in os.listdir(".") if n.endswith(".tsv")})
        ora_outputs, ora_files = _ORACLE_RUNS["c5"]
    finally:
        os.chdir(cwd)
    for name in ("c5.synteny_blocks.tsv", "c5.pre-collinear-merge.synteny_blocks.tsv"):
        assert eng.outputs[name] == ora_outputs[name], name
    mx = sorted(n for n in ora_files if ".k24.w1000.tsv" in n)
    assert len(mx) == 3
    for n in mx:
        assert hip_files[n] == ora_files[n], n
    # what the uniform family never reaches
    dups = 0
    for n in mx:
        hashes = np.array([int(tok.split(":")[0]) for line in ora_files[n].decode().splitlines() for tok in line.split("\t")[1].split(" ") if tok],
                          dtype=np.uint64)
        _, cnt = np.unique(hashes, return_counts=True)
        dups += int((cnt > 1).sum())
    print("assembly-like family:", eng.stats, "duplicated minimizer hashes within an assembly:", dups, seen)
    assert dups > 0, "no minimizer occurs twice within an assembly"
    assert seen["direct"] > 0, "no Bloom bucket overflowed / no lane in pieces"
    assert len(eng.outputs["c5.synteny_blocks.tsv"].splitlines()) // 3 >= 20
    assert seen["many"] > 0, "no select tile listed more candidates than its slots hold"
    assert eng.stats["small_blocks"] > 0 and eng.stats["merged"] > 0 and eng.stats["bubbles"] > 0 and eng.stats["indel_cuts"] > 0, eng.stats
