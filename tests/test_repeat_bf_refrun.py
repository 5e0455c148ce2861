"""The repeat filter (rule make_repeat_bf, experimental in the reference) against the reference's OWN script: tests/golden/repeat_bf/ holds
what bin/ntsynt_make_repeat_bfs.py's main() did with thirteen argument lists on two small families in the build container
(tests/golden/make_golden_repeat_bf.py: the script run over a stand-in btllib whose filter and hash rules are the restatement's).  Here,
without a GPU: oracle.nts_oracle.repeat_bf builds the same bits (size, popcount, SHA-1) for every run that finished; bin/ntsynt_make_repeat_bfs
of this build parses `--bf` and sizes the filter like the reference, and ends the refused argument lists with the same status and message.
tests/test_gpu_stages.py runs the HIP build on the same families."""
import contextlib
import gzip
import hashlib
import importlib.machinery
import importlib.util
import io
import json
import math
import os

import pytest

from oracle import nts_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
DIR = os.path.join(HERE, "golden", "repeat_bf")


@pytest.fixture(scope="module")
def vectors():
    with open(os.path.join(DIR, "cases.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="module")
def product():
    path = os.path.join(os.path.dirname(HERE), "bin", "ntsynt_make_repeat_bfs")
    loader = importlib.machinery.SourceFileLoader("product_make_repeat_bfs", path)
    spec = importlib.util.spec_from_loader("product_make_repeat_bfs", loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod


def unpack(dest, names):
    for f in names:
        with gzip.open(os.path.join(DIR, f + ".gz")) as fi, open(os.path.join(dest, f), "wb") as fo:
            fo.write(fi.read())


def options(argv):
    "the recorded argument list -> (genome files, k, --bf text or None, fpr)"
    files = argv[argv.index("--genome") + 1:]
    files = files[:next((i for i, a in enumerate(files) if a.startswith("-")), len(files))]
    get = lambda flag, default=None: argv[argv.index(flag) + 1] if flag in argv else default     # noqa: E731
    return files, int(get("-k")), get("--bf"), float(get("--fpr", "0.01"))


def test_the_restatement_builds_the_reference_runs_filters(vectors, product, tmp_path):
    unpack(str(tmp_path), vectors["families"]["a"] + vectors["families"]["b"])
    done = 0
    for case in vectors["cases"]:
        if case["end"] != "ran":
            continue
        files, k, bf_text, fpr = options(case["argv"])
        genomes = [O.read_fasta(str(tmp_path / f)) for f in files]
        out = io.StringIO()
        with contextlib.redirect_stdout(out):
            nbytes = product.parse_bf_size(bf_text, None) if bf_text else product.approximate_bf_size(genomes[0].total_bp, fpr)
        assert [ln for ln in out.getvalue().splitlines() if ln.startswith("Calculated")] == case["stdout"], case["argv"]
        if not bf_text:
            assert nbytes == int(math.ceil(-genomes[0].total_bp / math.log(1 - fpr)) / 8)
        rep = O.repeat_bf(genomes, k, O.bf_ctor_bytes(nbytes))
        assert (int(rep.size), int(O.bf_popcount(rep)), hashlib.sha1(rep.tobytes()).hexdigest()) == (case["bytes"], case["popcount"], case["sha1"]), case["argv"]
        done += 1
    assert done >= 7


def test_bf_size_units_like_the_reference(vectors, product):
    class Refuse:
        def print_help(self):
            pass

        def error(self, msg):
            raise ValueError(msg)
    for text, want in vectors["parse_bf_size"].items():
        try:
            got = product.parse_bf_size(text, Refuse())
        except ValueError as exc:
            got = "error: " + str(exc)
        assert got == want, text


def test_refused_argument_lists_end_like_the_reference(vectors, product):
    "(before any GPU work: the size is parsed first)"
    n = 0
    for case in vectors["cases"]:
        if case["end"] != "exit":
            continue
        so, se = io.StringIO(), io.StringIO()
        with contextlib.redirect_stdout(so), contextlib.redirect_stderr(se), pytest.raises(SystemExit) as exc:
            product.main(list(case["argv"]))
        assert exc.value.code == case["status"], case["argv"]
        err = [ln.split(": error: ", 1)[1] for ln in se.getvalue().splitlines() if ": error: " in ln]
        assert err and err[-1] == case["error"], case["argv"]
        assert ("usage:" in so.getvalue()) == case["printed_help"], case["argv"]
        n += 1
    assert n >= 5
