"""The Snakefile's command lines against this build's stage executables, without a GPU: tests/golden/smk_commands.json holds the shell lines
bin/ntsynt_run_pipeline.smk issues for fifteen configurations (tests/golden/make_golden_smk.py: the file's own Python executed and its own
`shell:` templates expanded in the build container, config = what bin/ntSynt hands over).  Every line must be ACCEPTED by the parser of the
executable it names (ntsynt_amd/stage_cli.py, bin/ntsynt_make_repeat_bfs) and must MEAN what the configuration says: the drop-in claim of
INTEGRATION.md section 1a ("the Snakefile's shell lines run against them unchanged") held against the Snakefile itself.
tests/test_gpu_stages.py runs one configuration's lines on the GPU."""
import importlib.machinery
import importlib.util
import json
import os
import shlex

import pytest
import yaml

from ntsynt_amd import stage_cli

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def runs(golden_dir):
    with open(os.path.join(golden_dir, "smk_commands.json")) as fh:
        return json.load(fh)["runs"]


@pytest.fixture(scope="module")
def repeat_parser():
    path = os.path.join(os.path.dirname(HERE), "bin", "ntsynt_make_repeat_bfs")
    loader = importlib.machinery.SourceFileLoader("product_make_repeat_bfs_p", path)
    spec = importlib.util.spec_from_loader("product_make_repeat_bfs_p", loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod.build_parser()


def test_every_line_of_the_snakefile_is_a_command_line_of_this_build(runs, repeat_parser):
    seen = {}
    for run in runs:
        cfg = {k: yaml.safe_load(v) for k, v in run["config"].items()}
        refs = cfg["references"]
        k, w, prefix = cfg["kmer"], cfg["window"], cfg["prefix"]
        for cmd in run["commands"]:
            words = shlex.split(cmd["shell"])
            rule = cmd["rule"]
            seen[rule] = seen.get(rule, 0) + 1
            if rule == "faidx":
                # samtools' job (out of scope); this build writes the .fai itself -- the line only has to name the files it names
                assert words[:3] == ["samtools", "faidx", "-o"] and words[3] == words[4] + ".fai"
            elif rule == "make_common_bf":
                assert words[0].endswith("/ntsynt_make_common_bf")
                a = stage_cli.make_common_bf_parser().parse_args(words[1:])
                assert (a.genome, a.k, a.fpr, a.p, a.t) == (refs, k, cfg["fpr"], f"{prefix}.common", cfg["threads"])
            elif rule == "make_repeat_bf":
                assert words[0].endswith("/ntsynt_make_repeat_bfs.py")
                a = repeat_parser.parse_args(words[1:])
                assert (a.genome, a.k, a.fpr, a.p, a.bf) == (refs, k, cfg["fpr"], f"{prefix}.repeat", None)
            elif rule == "indexlr":
                assert words[0] == "indexlr" and words[-2] == ">"
                a = stage_cli.indexlr_parser().parse_args(words[1:-2])
                fasta = cmd["wildcards"]["fasta"]
                assert (a.k, a.w, a.long, a.seq, a.pos, a.t) == (k, w, True, True, True, 5)
                assert a.s == (f"{prefix}.common.bf" if cfg["common"] is True else None)
                assert a.r == (f"{prefix}.repeat.bf" if cfg.get("repeat") is True else None)
                assert os.path.basename(a.fasta) == fasta and words[-1] == f"{fasta}.k{k}.w{w}.tsv"
            elif rule == "ntsynt_synteny":
                assert words[0] == "python3" and words[1].endswith("/ntsynt_run.py")
                a = stage_cli.run_parser().parse_args(words[2:])
                assert a.FILES == [f"{os.path.basename(r)}.k{k}.w{w}.tsv" for r in refs] and a.fastas == refs
                rounds = [int(x) for x in str(cfg["w_rounds"]).split()]
                assert (a.k, a.w, a.w_rounds, a.p, a.bp, a.z, a.btllib_t) == (k, w, rounds, prefix, cfg["indel_merge"], cfg["block_size"], cfg["threads"])
                assert a.collinear_merge == str(cfg["collinear_merge"])
                assert a.common == (f"{prefix}.common.bf" if cfg["common"] is True else None)
                assert a.simplify_graph == (cfg["simplify_graph"] is True) and a.dev == (cfg["dev"] is True)
                assert a.repeat == (f"{prefix}.repeat.bf" if cfg.get("repeat") is True else None)
                assert a.filter is None and a.n == 0 and a.m == 90                # (what the Snakefile never sets)
            else:
                raise AssertionError(f"a rule this build has no executable for: {rule}")
    assert set(seen) == {"faidx", "make_common_bf", "make_repeat_bf", "indexlr", "ntsynt_synteny"} and seen["ntsynt_synteny"] >= 10
