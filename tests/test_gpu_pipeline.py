"""End to end on the GPU: FASTA files -> byte-identical artefacts vs the CPU oracle pipeline, the
device graph build vs its numpy statement, and the ntSynt-compatible CLI."""
import os
import subprocess
import sys

import numpy as np
import pytest

from ntsynt_amd import synth
from oracle import nts_oracle as O
from oracle import synteny_oracle as SO
from tests.graph_ref import build_graph_numpy
from tests.helpers import oracle_flat

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _both(tmp_path, paths, **kw):
    from ntsynt_amd import pipeline
    cwd = os.getcwd()
    try:
        os.makedirs(tmp_path / "ora")
        os.makedirs(tmp_path / "hip")
        os.chdir(tmp_path / "ora")
        ora = SO.run_pipeline(paths, prefix="p", **kw)
        os.chdir(tmp_path / "hip")
        eng = pipeline.run(paths, prefix="p", log=lambda *a: None, **kw)
    finally:
        os.chdir(cwd)
    return ora, eng


CASES = [
    dict(n=3, bp=3_000_000, ctg=3, div=0.01, seed=21, micro=0, n_runs=True,
         kw=dict(k=24, w=1000, w_rounds=[100, 10], indel=500, merge=3000, block_size=500)),
    dict(n=2, bp=2_000_000, ctg=2, div=0.005, seed=9, micro=12, n_runs=False,
         kw=dict(k=24, w=200, w_rounds=[50, 10], indel=5000, merge=20000, block_size=300)),
    dict(n=4, bp=1_600_000, ctg=2, div=0.02, seed=33, micro=6, n_runs=True,
         kw=dict(k=20, w=500, w_rounds=[250, 100], indel=50000, merge="100w", block_size=1000)),
]


@pytest.mark.parametrize("case", CASES, ids=["d05-3g", "micro-2g", "d1-4g"])
def test_pipeline_byte_identical(tmp_path, case):
    paths = synth.make_family(str(tmp_path), case["n"], case["bp"], case["ctg"], case["div"], seed=case["seed"],
                              micro=case["micro"], n_runs=case["n_runs"], soft_mask=True,
                              line_width=(60 if case["seed"] % 2 else 0))
    ora, eng = _both(tmp_path, paths, **case["kw"])
    k, w = case["kw"]["k"], case["kw"]["w"]
    assert eng.outputs["p.synteny_blocks.tsv"] == ora.outputs["p.synteny_blocks.tsv"]
    assert eng.outputs["p.pre-collinear-merge.synteny_blocks.tsv"] == ora.outputs["p.pre-collinear-merge.synteny_blocks.tsv"]
    assert len(eng.outputs["p.synteny_blocks.tsv"].splitlines()) >= 2 * case["n"]
    for p in paths:
        name = f"{os.path.basename(p)}.k{k}.w{w}.tsv"
        assert open(tmp_path / "hip" / name).read() == open(tmp_path / "ora" / name).read(), name
    # the Bloom filter file holds the same bits as the oracle's cascade
    from ntsynt_amd.pipeline import read_bf
    bits, kk = read_bf(str(tmp_path / "hip" / "p.common.bf"))
    assert kk == k and np.array_equal(bits, ora.bf)


def test_no_paths_found_leaves_filter_and_minimizer_files_like_snakemake(tmp_path):
    """A family too distant for a chain of four common minimizers: both sides stop with the reference's "no paths found" (exit 1,
    bin/ntsynt_synteny.py:630-632).  Under Snakemake that fails rule ntsynt_synteny alone -- the filter file and the minimizer TSVs
    of the rules that had finished stay (complete: byte-identical to the oracle's), no block table is left behind."""
    from ntsynt_amd import pipeline
    paths = synth.make_family(str(tmp_path), 2, 400_000, 2, 0.35, seed=5)
    kw = dict(k=24, w=400, w_rounds=[100, 10], indel=500, merge=3000, block_size=300)
    names = [f"{os.path.basename(p)}.k24.w400.tsv" for p in paths]
    cwd = os.getcwd()
    texts = {}
    try:
        for side in ("ora", "hip"):
            os.makedirs(tmp_path / side)
            os.chdir(tmp_path / side)
            with pytest.raises(SystemExit) as e:
                if side == "ora":
                    SO.run_pipeline(paths, prefix="p", **kw)
                else:
                    pipeline.run(paths, prefix="p", log=lambda *a: None, **kw)
            assert e.value.code == 1
            files = set(os.listdir("."))
            assert not any(f.endswith("synteny_blocks.tsv") for f in files)
            texts[side] = [open(n).read() for n in names]
        assert texts["ora"] == texts["hip"]
        bits, k_file = pipeline.read_bf("p.common.bf")
        assert k_file == 24 and np.array_equal(bits, O.common_bf({p: O.read_fasta(p) for p in paths}, 24, 0.025, 1))
    finally:
        os.chdir(cwd)


def test_graph_build_device_vs_numpy(tmp_path):
    from ntsynt_amd.device import Context
    from ntsynt_amd.graph import build_graph_device
    paths = synth.make_family(str(tmp_path), 3, 1_500_000, 2, 0.01, seed=5, micro=8)
    genomes = [O.read_fasta(p) for p in paths]
    lists = [oracle_flat(O.minimize(g, 24, 300)) for g in genomes]
    rng = np.random.default_rng(3)
    keeps = [rng.random(len(l[0])) < 0.97 for l in lists]
    lids = [(l[1].astype(np.int64) * 1000 + np.cumsum(rng.random(len(l[0])) < 0.01)).astype(np.uint32) for l in lists]
    ctx = Context(0)
    for kp, li in ((None, None), (keeps, lids)):
        d = build_graph_device(ctx, lists, kp, li)
        h = build_graph_numpy(lists, kp, li)
        assert np.array_equal(d.v_hash, h.v_hash) and d.v_hash.size > 1000
        assert np.array_equal(d.occ_rec, h.occ_rec) and np.array_equal(d.occ_pos, h.occ_pos)
        # the device hands the edges over in ntJoin's dict-of-dicts order; the numpy twin in key order
        from ntsynt_amd.synteny import dict_order
        order = dict_order(h.e_u, h.e_first, h.v_hash.size)
        assert d.dict_ordered and not h.dict_ordered
        for f in ("e_u", "e_v", "e_w", "e_first"):
            assert np.array_equal(getattr(d, f), getattr(h, f)[order]), f
    ctx.close()


def test_cli_runs_like_the_reference_tests(tmp_path):
    "tests/ntsynt_tests.py:40-59 shape: `ntSynt ... -d 0.5 --indel 500 --merge 3000`, positional and --fastas_list"
    paths = synth.make_family(str(tmp_path), 3, 1_200_000, 2, 0.01, seed=8)
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, os.path.join(ROOT, "bin", "ntSynt"), "--force", *paths, "-k20", "-w", "500", "-d", "0.5",
           "--prefix", "cli", "--indel", "500", "--merge", "3000"]
    subprocess.run(cmd, cwd=tmp_path, env=env, check=True, stdout=subprocess.DEVNULL)
    fof = tmp_path / "list.tsv"
    fof.write_text("\n".join(paths) + "\n")
    cmd2 = [sys.executable, os.path.join(ROOT, "bin", "ntSynt"), "--force", "--fastas_list", str(fof), "-k20", "-w", "500",
            "-d", "0.5", "--prefix", "cli-fof", "--indel", "500", "--merge", "3000"]
    subprocess.run(cmd2, cwd=tmp_path, env=env, check=True, stdout=subprocess.DEVNULL)
    a = (tmp_path / "cli.synteny_blocks.tsv").read_text()
    b = (tmp_path / "cli-fof.synteny_blocks.tsv").read_text()
    assert a == b and len(a.splitlines()) >= 6
    cwd = os.getcwd()
    try:
        os.makedirs(tmp_path / "ora")
        os.chdir(tmp_path / "ora")
        ora = SO.run_pipeline(paths, k=20, w=500, w_rounds=[100, 10], indel=500, merge="3000", block_size=500, prefix="cli")
    finally:
        os.chdir(cwd)
    assert a == ora.outputs["cli.synteny_blocks.tsv"]
    for p in paths:
        assert os.path.exists(tmp_path / f"{os.path.basename(p)}.fai")


def test_pipeline_with_the_experimental_repeat_filter(tmp_path):
    """config "repeat" of the reference's Snakefile (rules make_repeat_bf, indexlr -r; ntsynt_run.py gets --repeat without --filter,
    so the refinement rounds do not use it): <prefix>.repeat.bf holds the oracle's bits, minimizer TSVs and synteny blocks are
    the oracle pipeline's."""
    from ntsynt_amd import pipeline, synth
    from oracle import nts_oracle as O
    from oracle import synteny_oracle as SO
    paths = synth.make_family(str(tmp_path), 3, 1_200_000, 4, 0.01, seed=44, micro=10)
    # give every genome something repeated: a copy of an earlier stretch further down the first record
    import re
    for p in paths:
        txt = open(p).read()
        recs = re.split(r"(?m)^>", txt)[1:]
        head, *lines = recs[0].split("\n")
        seq = "".join(lines)
        seq = seq[:200000] + seq[20000:60000] + seq[240000:]
        recs[0] = head + "\n" + "\n".join(seq[i:i + 80] for i in range(0, len(seq), 80)) + "\n"
        open(p, "w").write("".join(">" + r for r in recs))
    kw = dict(k=24, w=300, w_rounds=[100, 20], indel=400, merge="10w", block_size=300)
    cwd = os.getcwd()
    try:
        os.makedirs(tmp_path / "hip")
        os.makedirs(tmp_path / "ora")
        os.chdir(tmp_path / "hip")
        eng = pipeline.run(paths, prefix="p", log=lambda *a: None, repeat=True, **kw)
        os.chdir(tmp_path / "ora")
        ora = SO.run_pipeline(paths, prefix="p", repeat=True, **kw)
        os.makedirs(tmp_path / "plain")
        os.chdir(tmp_path / "plain")
        SO.run_pipeline(paths, prefix="p", **kw)
    finally:
        os.chdir(cwd)
    for name in ("p.synteny_blocks.tsv", "p.pre-collinear-merge.synteny_blocks.tsv"):
        assert eng.outputs[name] == ora.outputs[name], name
    differs = False
    for p in paths:
        tsv = f"{os.path.basename(p)}.k24.w300.tsv"
        assert open(tmp_path / "hip" / tsv).read() == open(tmp_path / "ora" / tsv).read()
        differs |= open(tmp_path / "ora" / tsv).read() != open(tmp_path / "plain" / tsv).read()
    assert differs                                       # the filter changes the sketches
    assert ora.outputs["p.synteny_blocks.tsv"] != "" and os.path.getsize(tmp_path / "hip" / "p.repeat.bf") > 1000
    genomes = [O.read_fasta(p) for p in paths]
    import math
    nb = (int(math.ceil(-genomes[0].total_bp / math.log(1 - 0.025)) / 8) + 7) // 8 * 8
    want = O.repeat_bf(genomes, 24, nb)
    raw = open(tmp_path / "hip" / "p.repeat.bf", "rb").read()
    assert raw[-nb:] == want.tobytes() and O.bf_popcount(want) > 10000


def test_benchmark_writes_stage_times_and_peak_memory(tmp_path):
    """`--benchmark` (bin/ntSynt:75; the reference wraps every rule in /usr/bin/time -v, smk:26-35, i.e. wall clock and peak RSS per
    stage): <prefix>.stage_times.tsv with the stages' seconds, the library's HBM high-water mark of the run and the process's peak
    RSS; the engine carries the same figures (bench.py's e2e leg reads them)."""
    import os
    from ntsynt_amd import pipeline, synth
    from ntsynt_amd.device import Context
    paths = synth.make_family(str(tmp_path), 2, 900_000, 2, 0.01, seed=8)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        eng = pipeline.run(paths, k=24, w=300, w_rounds=[100, 20], indel=500, merge=3000, block_size=300, prefix="b", benchmark=True, log=lambda *a: None)
    finally:
        os.chdir(cwd)
    rows = dict(line.rstrip("\n").split("\t") for line in open(tmp_path / "b.stage_times.tsv"))
    for stage in ("read_fasta+upload", "make_common_bf", "indexlr", "ntsynt_synteny", "wait_for_files"):
        assert float(rows[stage]) >= 0.0
    genome_bytes = sum(os.path.getsize(p) for p in paths)
    assert int(rows["peak_hbm_bytes"]) > genome_bytes and int(rows["peak_host_rss_bytes"]) > 50 << 20
    assert eng.memory["peak_hbm_bytes"] == int(rows["peak_hbm_bytes"]) and "bf_first_insert" in eng.memory["hbm_live_at_marks"]
    # the counter itself: live bytes go up and down with allocations, the mark only up, a reset brings it down to what is live
    ctx = Context(0)
    before = ctx.mem_stats()
    from ntsynt_amd.device import BloomFilter
    bf = BloomFilter(ctx, 64 << 20, 24)
    mid = ctx.mem_stats()
    assert mid["live"] >= before["live"] + (64 << 20) and mid["peak"] >= mid["live"] and mid["device_total"] > mid["device_used"] > 0
    bf.free()
    after = ctx.mem_stats()
    assert after["live"] <= mid["live"] - (64 << 20) and after["peak"] == mid["peak"]
    ctx.mem_reset_peak()
    assert ctx.mem_stats()["peak"] == ctx.mem_stats()["live"]
    ctx.close()
