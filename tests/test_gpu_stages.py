"""The stage executables of the reference's workflow (bin/ntsynt_make_common_bf, bin/indexlr, bin/ntsynt_run.py: the command
lines of rules make_common_bf / indexlr / ntsynt_synteny, bin/ntsynt_run_pipeline.smk:55-103) chained through files like the
Snakefile chains them, against `ntSynt` in one go -- and the reference's own minimizer files through the stage-3 command line."""
import os
import subprocess
import sys

import numpy as np
import pytest

from ntsynt_amd import synth
from oracle import synteny_oracle as SO

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bin")


def _run(cmd, cwd, stdout=None):
    r = subprocess.run([sys.executable] + cmd, cwd=cwd, stdout=stdout or subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, (cmd, r.stderr.decode()[-2000:])
    return r


def test_stages_chained_through_files_equal_ntsynt_in_one_go(tmp_path):
    """make_common_bf -> indexlr x 3 -> ntsynt_run.py with the argument strings of smk:62,85,102-103, each a process of its own,
    data crossing as files in the CWD -- every artefact byte-identical to bin/ntSynt's (filter file, minimizer TSVs, both synteny
    TSVs), and both equal to the oracle pipeline's."""
    src = tmp_path / "in"
    src.mkdir()
    paths = synth.make_family(str(src), 3, 900_000, 3, 0.01, seed=41, micro=8, soft_mask=True, n_runs=True, line_width=70)
    one, chain = tmp_path / "one", tmp_path / "chain"
    one.mkdir()
    chain.mkdir()
    k, w = 24, 400
    opts = ["-k", str(k), "-w", str(w), "--w_rounds", "100", "20", "--indel", "600", "--merge", "3000", "-b", "300"]
    _run([os.path.join(BIN, "ntSynt")] + paths + ["-d", "1", "-p", "run"] + opts, str(one))
    # rule make_common_bf (smk:62): {script} --genome {refs} -p {prefix}.common --fpr {fpr} -k {k} -t {threads}
    out = _run([os.path.join(BIN, "ntsynt_make_common_bf"), "--genome"] + paths + ["-p", "run.common", "--fpr", "0.025", "-k", str(k), "-t", "12"], str(chain))
    text = out.stdout.decode()
    assert "BF size (bytes):" in text and text.count("Bloom filter FPR:") == 4          # three levels + the final line (cpp:132,154,162)
    # rule indexlr (smk:85): indexlr -k -w --long --seq --pos -t 5 -s {common} {fa} > {fa}.k{k}.w{w}.tsv
    tsvs = []
    for p in paths:
        t = f"{os.path.basename(p)}.k{k}.w{w}.tsv"
        with open(chain / t, "wb") as fh:
            _run([os.path.join(BIN, "indexlr"), "-k", str(k), "-w", str(w), "--long", "--seq", "--pos", "-t", "5", "-s", "run.common.bf", p], str(chain), stdout=fh)
        tsvs.append(t)
    # rule ntsynt_synteny (smk:102-103)
    out = _run([os.path.join(BIN, "ntsynt_run.py")] + tsvs + ["-k", str(k), "-w", str(w), "--w-rounds", "100", "20", "-p", "run", "--bp", "600",
                                                             "--collinear-merge", "3000", "-z", "300", "--common", "run.common.bf", "--simplify-graph",
                                                             "--btllib_t", "12", "--fastas"] + paths, str(chain))
    assert "Running ntSynt v1.0.4" in out.stdout.decode()
    names = ["run.common.bf", "run.synteny_blocks.tsv", "run.pre-collinear-merge.synteny_blocks.tsv"] + tsvs
    for n in names:
        a, b = open(one / n, "rb").read(), open(chain / n, "rb").read()
        assert a == b and len(a) > 0, n
    assert not [n for n in os.listdir(chain) if n.endswith(".fai")]                  # stage 3 writes no index (rule faidx is its own)
    cwd = os.getcwd()
    os.makedirs(tmp_path / "ora")
    os.chdir(tmp_path / "ora")
    try:
        ora = SO.run_pipeline(paths, k=k, w=w, w_rounds=[100, 20], indel=600, merge=3000, block_size=300, prefix="run", threads=4)
    finally:
        os.chdir(cwd)
    for n in ("run.synteny_blocks.tsv", "run.pre-collinear-merge.synteny_blocks.tsv"):
        assert open(chain / n).read() == ora.outputs[n], n
    assert len(ora.outputs["run.synteny_blocks.tsv"].splitlines()) >= 9


MX = {("ref", 24): "mx_celegans-chrII-III.fa.k24.w1000.npz", ("A", 24): "mx_celegans-chrII-III.A.fa.k24.w1000.npz",
      ("ref", 20): "mx_celegans-chrII-III.fa.k20.w1000.npz", ("A", 20): "mx_celegans-chrII-III.A.fa.k20.w1000.npz",
      ("B", 20): "mx_celegans-chrII-III.B.fa.k20.w1000.npz"}
FASTA = {"ref": "celegans-chrII-III.fa", "A": "celegans-chrII-III.A.fa", "B": "celegans-chrII-III.B.fa"}


@pytest.mark.parametrize("keys,k,n_blocks", [([("ref", 24), ("A", 24)], 24, 15), ([("ref", 20), ("A", 20), ("B", 20)], 20, 16)])
def test_reference_minimizer_files_through_the_stage3_command_line(tmp_path, golden_dir, keys, k, n_blocks):
    """The reference's own indexlr output (tests/expected_result/*.k{20,24}.w1000.tsv, committed as arrays) written back as
    minimizer TSVs and given to bin/ntsynt_run.py: the initial round's table (the one the reference overwrites at S:516-523,
    so the FASTA files -- absent from the reference tree -- are not needed) equals the oracle's on the same lists, with the
    block counts of SURVEY.md 8(c) P4 (15 and 16 with bubble removal, the workflow's default)."""
    tsvs, tables = [], {}
    for key in keys:
        z = np.load(os.path.join(golden_dir, MX[key]))
        names = [str(c) for c in z["contigs"]]
        name = f"{FASTA[key[0]]}.k{k}.w1000.tsv"
        recs = [(c, []) for c in names]
        for ci, h, p in zip(z["contig_idx"].tolist(), z["h1"].tolist(), z["pos"].tolist()):
            recs[ci][1].append((str(h), p))
        with open(tmp_path / name, "w") as fh:
            for c, toks in recs:
                fh.write(c + "\t" + " ".join(f"{h}:{p}" for h, p in toks) + "\n")
        tsvs.append(name)
        tables[name] = SO.mx_tables_from_tokens(recs)
    fastas = [FASTA[key[0]] for key in keys]                                     # (names only: --initial-only reads no sequence)
    _run([os.path.join(BIN, "ntsynt_run.py")] + tsvs + ["-k", str(k), "-w", "1000", "--w-rounds", "100", "10", "-p", "c1", "--bp", "500",
                                                       "--collinear-merge", "3000", "-z", "500", "--simplify-graph", "--initial-only", "--fastas"] + fastas,
         str(tmp_path))
    got = open(tmp_path / "c1.synteny_blocks.tsv").read()
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        ora = SO.SyntenyOracle(list(tables), {}, k, 1000, [], 500, 3000, 500, "ora", simplify=True)
        ora.load(tables)
        want = ora.main()["ora.synteny_blocks.tsv"]
    finally:
        os.chdir(cwd)
    assert got == want and len(got.splitlines()) == n_blocks * len(keys)


def test_stage_executables_options(tmp_path):
    """--bf (explicit filter size, cpp:109-113), a gzip FASTA through indexlr with -o, indexlr without a filter, and stage 3
    without --common (refinement rounds sketch unfiltered) -- each against the oracle on the same inputs"""
    import gzip
    import shutil
    from oracle import nts_oracle as O
    from ntsynt_amd.pipeline import read_bf
    src = tmp_path / "in"
    src.mkdir()
    paths = synth.make_family(str(src), 2, 700_000, 2, 0.01, seed=52, micro=5, soft_mask=True)
    k, w = 20, 300
    # make_common_bf --bf: the size as given (+ the constructor's rounding), bits = the oracle's cascade at that size
    _run([os.path.join(BIN, "ntsynt_make_common_bf"), "--genome", paths[1], paths[0], "-p", "sized", "-k", str(k), "--bf", "777777"], str(tmp_path))
    bits, k_file = read_bf(str(tmp_path / "sized.bf"))
    genomes = {p: O.read_fasta(p) for p in paths}
    nbytes = O.bf_ctor_bytes(777777)
    want = O.bf_build(genomes[sorted(paths)[1]], k, nbytes, prev=O.bf_build(genomes[sorted(paths)[0]], k, nbytes))
    assert k_file == k and bits.size == nbytes and np.array_equal(bits, want)
    # indexlr: gzip input, -o, with and without the filter
    gz = str(tmp_path / "a.fa.gz")
    with open(paths[0], "rb") as fi, gzip.open(gz, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    for label, extra, bf in (("filtered", ["-s", "sized.bf"], want), ("plain", [], None)):
        out = tmp_path / f"{label}.tsv"
        _run([os.path.join(BIN, "indexlr"), "-k", str(k), "-w", str(w), "--long", "--seq", "--pos", "-t", "5"] + extra + ["-o", str(out), gz], str(tmp_path))
        ref = tmp_path / f"{label}.ora.tsv"
        O.write_indexlr_tsv(str(ref), genomes[paths[0]], O.minimize(genomes[paths[0]], k, w, bf), k)
        assert out.read_bytes() == ref.read_bytes() and out.stat().st_size > 1000, label
    # stage 3 without --common: unfiltered initial lists in, unfiltered refinement sketches
    tsvs = []
    for p in paths:
        t = f"{os.path.basename(p)}.k{k}.w{w}.tsv"
        O.write_indexlr_tsv(str(tmp_path / t), genomes[p], O.minimize(genomes[p], k, w, None), k)
        tsvs.append(t)
    _run([os.path.join(BIN, "ntsynt_run.py")] + tsvs + ["-k", str(k), "-w", str(w), "--w-rounds", "100", "20", "-p", "nc", "--bp", "500", "--collinear-merge", "2w",
                                                       "-z", "300", "--fastas"] + paths, str(tmp_path))
    cwd = os.getcwd()
    os.makedirs(tmp_path / "ora")
    os.chdir(tmp_path / "ora")
    try:
        ora = SO.run_pipeline(paths, k=k, w=w, w_rounds=[100, 20], indel=500, merge="2w", block_size=300, prefix="nc", common=False, simplify=False)
    finally:
        os.chdir(cwd)
    for n in ("nc.synteny_blocks.tsv", "nc.pre-collinear-merge.synteny_blocks.tsv"):
        assert open(tmp_path / n).read() == ora.outputs[n], n


def test_stage3_filter_indexlr_and_filter_filter_against_the_oracle_engine(tmp_path):
    """`ntsynt_run.py --filter Indexlr --repeat <bf> --common <bf>` (bin/ntsynt_synteny.py:172-180, 598-599): the refinement rounds'
    indexlr runs get `-r <repeat filter>` next to `-s <common filter>`; the initial lists are read from the files as they are.
    Against the oracle engine with the same filters; the repeat filter must change the result (else the test proves nothing);
    `--filter Filter --repeat <bf>` (S:183-184,601-604): ntJoin's read_minimizers(file, repeat_bf) leaves out the minimizers whose k-mer
    the filter holds -- initial files and refinement lists (nts_mx_screen on the device; the oracle screens the TSV's k-mer text).
    `--filter` without `--repeat` is the reference's ValueError."""
    import re
    from oracle import nts_oracle as O
    from ntsynt_amd.pipeline import write_bf
    src = tmp_path / "in"
    src.mkdir()
    paths = synth.make_family(str(src), 3, 900_000, 3, 0.01, seed=61, micro=8)
    for p in paths:                                         # something repeated in every genome: a stretch of the first record copied further down
        recs = re.split(r"(?m)^>", open(p).read())[1:]
        head, *lines = recs[0].split("\n")
        seq = "".join(lines)
        seq = seq[:150000] + seq[10000:70000] + seq[150000:]
        recs[0] = head + "\n" + "\n".join(seq[i:i + 80] for i in range(0, len(seq), 80)) + "\n"
        open(p, "w").write("".join(">" + r for r in recs))
    k, w = 24, 400
    genomes = {p: O.read_fasta(p) for p in paths}
    common = O.common_bf(genomes, k, 0.025, 1)
    rep = O.repeat_bf([genomes[p] for p in paths], k, common.size)
    assert O.bf_popcount(rep) > 10000
    write_bf(str(tmp_path / "f.common.bf"), common, k)
    write_bf(str(tmp_path / "f.repeat.bf"), rep, k)
    tsvs, tables, by_tsv = [], {}, {}
    for p in paths:
        t = f"{os.path.basename(p)}.k{k}.w{w}.tsv"
        O.write_indexlr_tsv(str(tmp_path / t), genomes[p], O.minimize(genomes[p], k, w, common), k)
        tsvs.append(t)
        tables[t] = SO.read_minimizers_tsv(str(tmp_path / t))
        by_tsv[t] = genomes[p]
    args = tsvs + ["-k", str(k), "-w", str(w), "--w-rounds", "100", "20", "--bp", "600", "--collinear-merge", "3000", "-z", "300", "--common", "f.common.bf",
                   "--simplify-graph", "--fastas"] + paths
    _run([os.path.join(BIN, "ntsynt_run.py")] + args + ["-p", "withr", "--filter", "Indexlr", "--repeat", "f.repeat.bf"], str(tmp_path))
    _run([os.path.join(BIN, "ntsynt_run.py")] + args + ["-p", "plain", "--repeat", "f.repeat.bf"], str(tmp_path))     # (--repeat alone: echoed, unused)
    # --filter Filter: the minimizers whose k-mer the repeat filter holds are not read (initial files and every refinement round's lists)
    _run([os.path.join(BIN, "ntsynt_run.py")] + args + ["-p", "screened", "--filter", "Filter", "--repeat", "f.repeat.bf"], str(tmp_path))
    outs = {}
    cwd = os.getcwd()
    for label, r in (("withr", rep), ("plain", None), ("screened", rep)):
        os.makedirs(tmp_path / f"ora_{label}")
        os.chdir(tmp_path / f"ora_{label}")
        try:
            eng = SO.SyntenyOracle(list(tables), by_tsv, k, w, [100, 20], 600, 3000, 300, label, bf=common, simplify=True)
            if label == "screened":
                eng.screen_repeat = r
                eng.load({t: SO.read_minimizers_tsv(str(tmp_path / t), repeat_bf=r) for t in tsvs})
            else:
                eng.refine_repeat = r
                eng.load(tables)
            eng.main()
        finally:
            os.chdir(cwd)
        outs[label] = eng.outputs
    assert outs["screened"]["screened.synteny_blocks.tsv"] not in (outs["plain"]["plain.synteny_blocks.tsv"], outs["withr"]["withr.synteny_blocks.tsv"])
    for label in ("withr", "plain", "screened"):
        for n in (f"{label}.synteny_blocks.tsv", f"{label}.pre-collinear-merge.synteny_blocks.tsv"):
            assert open(tmp_path / n).read() == outs[label][n] and outs[label][n], n
    assert outs["withr"]["withr.synteny_blocks.tsv"] != outs["plain"]["plain.synteny_blocks.tsv"]
    r = subprocess.run([sys.executable, os.path.join(BIN, "ntsynt_run.py")] + args + ["-p", "x", "--filter", "Indexlr"], cwd=str(tmp_path), capture_output=True)
    assert r.returncode != 0 and b"must supply repeat Bloom filter with --repeat" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(BIN, "ntsynt_run.py")] + args + ["-p", "x", "--filter", "Filter", "--repeat", "f.repeat.bf", "--initial-only"],
                       cwd=str(tmp_path), capture_output=True)
    assert r.returncode == 2 and b"not with --initial-only" in r.stderr


def test_repeat_filter_executable_writes_the_reference_runs_bits(tmp_path):
    """bin/ntsynt_make_repeat_bfs (nts_bf_insert_repeats) on the families of tests/golden/repeat_bf/: the bits of the file it writes are the bits
    the reference's own bin/ntsynt_make_repeat_bfs.py left in its filter when it was run on the same arguments in the build container
    (tests/golden/make_golden_repeat_bf.py; stand-in btllib with the restatement's filter and hash rules)."""
    import gzip
    import hashlib
    import json
    from ntsynt_amd.pipeline import read_bf
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "repeat_bf")
    with open(os.path.join(src, "cases.json")) as fh:
        vec = json.load(fh)
    for f in vec["families"]["a"] + vec["families"]["b"]:
        with gzip.open(os.path.join(src, f + ".gz")) as fi, open(tmp_path / f, "wb") as fo:
            fo.write(fi.read())
    done = 0
    for case in vec["cases"]:
        if case["end"] != "ran":
            continue
        out = _run([os.path.join(BIN, "ntsynt_make_repeat_bfs")] + case["argv"], str(tmp_path)).stdout.decode()
        assert [ln for ln in out.splitlines() if ln.startswith("Calculated")] == case["stdout"], case["argv"]
        bits, k = read_bf(str(tmp_path / case["saved"]))
        assert k == int(case["argv"][case["argv"].index("-k") + 1])
        assert (int(bits.size), hashlib.sha1(bits.tobytes()).hexdigest()) == (case["bytes"], case["sha1"]), case["argv"]
        os.remove(tmp_path / case["saved"])
        done += 1
    assert done >= 7


def test_the_snakefiles_own_command_lines_run_on_this_builds_executables(tmp_path, golden_dir):
    """One configuration of tests/golden/smk_commands.json (the lines bin/ntsynt_run_pipeline.smk issues for `ntSynt -d 3 a.fa b.fa`, expanded
    from the Snakefile itself in the build container) run word for word in the rule order Snakemake would take -- make_common_bf, indexlr per
    genome, ntsynt_synteny; samtools' faidx lines are not this build's -- with <bin> = this build's bin/.  The final table against the CPU
    restatement with the configuration's parameters."""
    import json
    import shlex
    import yaml
    with open(os.path.join(golden_dir, "smk_commands.json")) as fh:
        runs = json.load(fh)["runs"]
    run = next(r for r in runs if r["config"]["w_rounds"] == "250 100" and r["config"]["common"] == "True" and r["config"]["references"] == "[a.fa, b.fa]"
               and r["config"]["prefix"] == "ntSynt.k24.w1000" and r["config"]["indel_merge"] == "50000")
    cfg = {k: yaml.safe_load(v) for k, v in run["config"].items()}
    paths = synth.make_family(str(tmp_path), 2, 2_500_000, 3, 0.01, seed=77, micro=6)
    for p, name in zip(paths, cfg["references"]):
        os.rename(p, tmp_path / name)
    order = {"make_common_bf": 0, "indexlr": 1, "ntsynt_synteny": 2}
    for cmd in sorted((c for c in run["commands"] if c["rule"] in order), key=lambda c: order[c["rule"]]):
        words = shlex.split(cmd["shell"])
        out = None
        if ">" in words:
            out = words[words.index(">") + 1]
            words = words[:words.index(">")]
        if words[0] == "python3":
            words = words[1:]
        words[0] = os.path.join(BIN, os.path.basename(words[0]))
        r = subprocess.run([sys.executable] + words, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, (cmd["shell"], r.stderr.decode()[-1500:])
        if out:
            with open(tmp_path / out, "wb") as fh:
                fh.write(r.stdout)
        for f in cmd["output"]:
            assert os.path.exists(tmp_path / f), (cmd["rule"], f)
    cwd = os.getcwd()
    os.makedirs(tmp_path / "ora")
    os.chdir(tmp_path / "ora")
    try:
        ora = SO.run_pipeline([str(tmp_path / n) for n in cfg["references"]], k=cfg["kmer"], w=cfg["window"], fpr=cfg["fpr"],
                              w_rounds=[int(x) for x in cfg["w_rounds"].split()], indel=cfg["indel_merge"], merge=cfg["collinear_merge"],
                              block_size=cfg["block_size"], prefix=cfg["prefix"])
    finally:
        os.chdir(cwd)
    for n in (f"{cfg['prefix']}.synteny_blocks.tsv", f"{cfg['prefix']}.pre-collinear-merge.synteny_blocks.tsv"):
        assert open(tmp_path / n).read() == ora.outputs[n], n
    assert len(ora.outputs[f"{cfg['prefix']}.synteny_blocks.tsv"].splitlines()) >= 4
