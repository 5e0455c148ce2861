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
def test_pipeline_on_a_structural_family_matches_the_oracle(ctx, tmp_path, monkeypatch, erode_on_host):
    """three genomes with inversions, translocations, indels and soft-masked stretches, written to FASTA from HBM like
    bench.py's e2e leg does; both synteny TSVs byte-identical to the oracle pipeline's, and the rules have fired"""
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
