#!/usr/bin/env python3
"""Command-line vectors made by RUNNING the reference's two Python entry points -- build container only (/root/reference):

  * bin/ntSynt's main() (bin/ntSynt:33-170), imported as a module and run on argument lists drawn to reach every branch of its
    argument handling: the divergence table with its `or` defaults (a 0 given on the command line is replaced: `args.indel or 10000`),
    a negative divergence (accepted: `< 1`), the w_rounds check (`>`: a round equal to -w passes), positional files / --fastas_list /
    both / neither / fewer than two, a missing input file, the hidden switches, -n and -f.  What is kept per case: how the call ended
    (argparse's exit status and message, the exception's text, or the `snakemake ... --config key=value` line it hands to
    subprocess.call, parsed into its keys) and the "Parameter settings" lines it prints.
  * bin/ntsynt_run.py's parse_arguments() (bin/ntsynt_run.py:10-44): the namespace stage 3 is constructed from.

Stand-ins (the image has neither): a module `snakemake` with __version__ = "7.32.4" (so `--rerun-trigger mtime` is appended, as under
any Snakemake since 7.8.0 -- the product has no Snakemake to hand it to), subprocess.call replaced by a recorder that returns 0, and
for ntsynt_run.py an empty `ntsynt_synteny` module (only the parser runs).  tests/test_cli_refrun.py holds ntsynt_amd/cli.py and
ntsynt_amd/stage_cli.py against these vectors.  No reference source text is stored: argument lists in, outcomes out.

  python tests/golden/make_golden_cli.py"""
import contextlib
import importlib.machinery
import importlib.util
import io
import json
import os
import random
import sys
import tempfile
import types

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load(path, name, stubs):
    for mod_name, mod in stubs.items():
        sys.modules[mod_name] = mod
    loader = importlib.machinery.SourceFileLoader(name, path)
    spec = importlib.util.spec_from_loader(name, loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod


def last_error(stderr):
    "argparse's `prog: error: message` line, without the program name"
    for line in reversed(stderr.splitlines()):
        if ": error: " in line:
            return line.split(": error: ", 1)[1]
    return None


def parse_command(words):
    "the recorded snakemake command -> {config key: value}, flags"
    assert words[0] == "snakemake"
    at = words.index("--config")
    cfg, rest = {}, []
    for wd in words[at + 1:]:
        if "=" in wd and not wd.startswith("-") and not rest:
            key, val = wd.split("=", 1)
            cfg[key] = val
        else:
            rest.append(wd)
    return {"cores": words[words.index("--cores") + 1], "config": cfg, "dry_run": "-n" in rest, "force": "-F" in rest,
            "load": rest[rest.index("--resources") + 1] if "--resources" in rest else None}


def run_ntsynt(mod, argv, cwd):
    calls = []
    so, se = io.StringIO(), io.StringIO()
    here = os.getcwd()
    os.chdir(cwd)
    old_argv, old_call = sys.argv, mod.subprocess.call
    sys.argv = ["ntSynt"] + argv
    mod.subprocess.call = lambda cmd, *a, **k: calls.append(cmd) or 0
    out = {}
    try:
        with contextlib.redirect_stdout(so), contextlib.redirect_stderr(se):
            try:
                mod.main()
                out["end"] = "ran"
            except SystemExit as exc:
                out["end"] = "exit"
                out["status"] = exc.code
                out["error"] = last_error(se.getvalue())
            except Exception as exc:                             # noqa: BLE001 -- the reference's own (FileNotFoundError)
                out["end"] = "raised"
                out["exception"] = f"{type(exc).__name__}: {exc}"
    finally:
        sys.argv, mod.subprocess.call = old_argv, old_call
        os.chdir(here)
    text = so.getvalue()
    if "Parameter settings:" in text:
        lines = text.splitlines()
        at = lines.index("Parameter settings:")
        out["settings"] = [ln for ln in lines[at + 1:at + 12] if ln.startswith("\t")]
    if calls:
        out["command"] = parse_command(calls[0])
    return out


def ntsynt_cases(rng):
    two, three = ["a.fa", "b.fa"], ["a.fa", "b.fa", "c.fa"]
    cases = []

    def add(argv, note=""):
        cases.append({"argv": argv, "note": note})
    for d in ("0", "0.1", "0.99", "1", "1.0", "5", "10", "10.0", "10.0001", "50", "100", "100.5", "-1", "-0.5", "1e-3", "nan"):
        add(["-d", d] + two, "divergence class")
    add(["-d=-1"] + two, "negative divergence is below 1")
    for flag, vals in (("--indel", ["0", "1", "77"]), ("--merge", ["0", "5000", "3w", "0w", "w"]), ("-b", ["0", "1", "250"]),
                       ("--block_size", ["123"])):
        for v in vals:
            for d in ("0.5", "3", "20"):
                add(["-d", d, flag, v] + two, "explicit value / falsy value replaced by the table's")
    for d in ("0.5", "3", "20"):
        add(["-d", d, "--w_rounds", "50", "5", "--"] + two)
        add(["-d", d] + two + ["--w_rounds", "1000"], "a round equal to -w passes the > test")
        add(["-d", d] + two + ["--w_rounds", "1001"], "a round above -w")
        add(["-d", d, "-w", "200"] + two, "the table's rounds against a smaller -w")
        add(["-d", d, "-w", "250"] + two)
        add(["-d", d, "-w", "99"] + two)
        add(["-d", d] + two + ["--w_rounds", "100", "100"], "duplicates pass here (stage 3 refuses them, S:597-599)")
        add(["-d", d] + two + ["--w_rounds", "10", "100"], "increasing rounds pass")
    add(["-d", "1"], "no input")
    add(["-d", "1", "a.fa"], "one genome")
    add(["-d", "1", "--fastas_list", "list2.txt"])
    add(["-d", "1", "--fastas_list", "list3_blank.txt"], "a blank line is a file name of its own: ''")
    add(["-d", "1", "--fastas_list", "list1.txt"], "one genome listed")
    add(["-d", "1", "--fastas_list", "list_empty.txt"], "an empty list is as good as none for the first test... and then too short")
    add(["-d", "1", "--fastas_list", "list2.txt", "a.fa"], "both")
    add(["-d", "1", "--fastas_list", "nowhere.txt"], "the list itself is missing")
    add(["-d", "1", "a.fa", "missing.fa"], "an input that does not exist")
    add(["-d", "1", "--fastas_list", "list_missing.txt"])
    add(two, "-d is required")
    add(["-d"] + two, "-d takes a number")
    add(["-d", "1", "-k", "abc"] + two)
    add(["-d", "1", "--frobnicate"] + two)
    add(["-d", "1", "-k", "32", "-w", "500"] + three, "prefix from k and w")
    add(["-d", "1", "-p", "run7", "-k", "20"] + three)
    add(["-d", "1", "--prefix", "x/y"] + two)
    add(["-d", "1", "-t", "48", "--fpr", "0.001"] + two)
    add(["-d", "1", "--fpr", "1e-4"] + two)
    for sw in (["--no-common"], ["--no-simplify-graph"], ["--benchmark"], ["--dev"], ["-n"], ["--dry-run"], ["-f"], ["--force"],
               ["--no-common", "--no-simplify-graph", "--benchmark", "--dev", "-n", "-f"]):
        add(["-d", "2"] + sw + two)
    add(["--divergence", "0.2", "--block_size", "2000", "--merge", "2w", "--indel", "300", "--w_rounds", "64", "8", "-w", "64", "-k", "16"] + three)
    flags = [["--indel"], ["--merge"], ["-b"], ["--w_rounds"], ["-w"], ["-k"], ["-p"], ["-t"], ["--fpr"]]
    for _ in range(60):
        argv = ["-d", rng.choice(["0.3", "1", "7.5", "10", "33", "100"])]
        for fl in rng.sample(flags, rng.randint(0, 5)):
            name = fl[0]
            if name == "--w_rounds":
                argv += [name] + [str(rng.choice([2000, 1000, 500, 250, 100, 33, 10, 4])) for _ in range(rng.randint(1, 3))]
            elif name == "--merge":
                argv += [name, rng.choice(["0", "1", "20000", "1w", "10w", "2.5w"])]
            elif name == "-p":
                argv += [name, rng.choice(["out", "p.q", "ntSynt.k24.w1000"])]
            elif name == "--fpr":
                argv += [name, rng.choice(["0.025", "0.1", "0.5"])]
            elif name == "-w":
                argv += [name, rng.choice(["1000", "300", "64", "5000"])]
            elif name == "-k":
                argv += [name, rng.choice(["16", "24", "31", "64"])]
            else:
                argv += [name, str(rng.choice([0, 1, 50, 999, 100000]))]
        argv += rng.choice([[], ["--no-common"], ["--dev"], ["-n"]])
        files = rng.choice([two, three, three + ["d.fa.gz"]])
        argv = argv + ["--"] + files if rng.random() < 0.5 else files + argv
        add(argv, "random")
    return cases


def run_cases():
    rng = random.Random(20260930)
    snk = types.ModuleType("snakemake")
    snk.__version__ = "7.32.4"
    mod = load(os.path.join(REF, "bin", "ntSynt"), "ref_ntsynt_cli", {"snakemake": snk})
    files = {"a.fa": ">a\nACGT\n", "b.fa": ">b\nACGT\n", "c.fa": ">c\nACGT\n", "d.fa.gz": "", "list2.txt": "a.fa\nb.fa\n",
             "list3_blank.txt": "a.fa\n\nb.fa\n", "list1.txt": "a.fa\n", "list_empty.txt": "", "list_missing.txt": "a.fa\nmissing.fa\n"}
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        for name, text in files.items():
            with open(os.path.join(tmp, name), "w") as fh:
                fh.write(text)
        for case in ntsynt_cases(rng):
            res = run_ntsynt(mod, case["argv"], tmp)
            out.append(dict(case, **res))
    return {"files": files, "cases": out}


def stage3_cases():
    "bin/ntsynt_run.py's parser: the namespace NtSyntSynteny(args) is built from"
    sys.path.insert(0, os.path.join(REF, "bin"))
    stub = types.ModuleType("ntsynt_synteny")
    stub.NtSyntSynteny = object
    mod = load(os.path.join(REF, "bin", "ntsynt_run.py"), "ref_ntsynt_run", {"ntsynt_synteny": stub})
    base = ["a.fa.k24.w1000.tsv", "b.fa.k24.w1000.tsv", "--fastas", "a.fa", "b.fa", "-k", "24", "-w", "1000"]
    argvs = [base,
             base + ["-n", "2", "-p", "pre", "-z", "1000", "--w-rounds", "250", "100", "--bp", "50000", "--collinear-merge", "100000", "--simplify-graph",
                     "--common", "pre.common.bf", "--btllib_t", "12"],
             base + ["--collinear-merge", "3w", "-m", "75", "--dev", "--interarrivals"],
             base + ["--filter", "Indexlr", "--repeat", "r.bf"],
             base + ["--filter", "Filter", "--repeat", "r.bf"],
             base + ["--filter", "other"],
             base + ["--w-rounds"],
             base[:-2],                                            # -w is required
             base[:2] + ["-k", "24", "-w", "1000"],                # --fastas is required
             ["--fastas", "a.fa", "b.fa", "-k", "24", "-w", "1000"],   # no minimizer file
             base + ["-z", "x"],
             base + ["--bp", "0", "-z", "0", "-n", "0"]]
    out = []
    for argv in argvs:
        so, se = io.StringIO(), io.StringIO()
        old = sys.argv
        sys.argv = ["ntsynt_run.py"] + argv
        rec = {"argv": argv}
        try:
            with contextlib.redirect_stdout(so), contextlib.redirect_stderr(se):
                try:
                    ns = mod.parse_arguments()
                    rec["end"] = "parsed"
                    rec["namespace"] = dict(sorted(vars(ns).items()))
                except SystemExit as exc:
                    rec["end"] = "exit"
                    rec["status"] = exc.code
                    rec["error"] = last_error(se.getvalue())
        finally:
            sys.argv = old
        out.append(rec)
    return out


def main():
    data = {"ntSynt": run_cases(), "ntsynt_run": stage3_cases(), "snakemake_version_assumed": "7.32.4"}
    with open(os.path.join(OUT, "cli_cases.json"), "w") as fh:
        json.dump(data, fh, separators=(",", ":"), sort_keys=True)
        fh.write("\n")
    ends = {}
    for c in data["ntSynt"]["cases"]:
        ends[c["end"]] = ends.get(c["end"], 0) + 1
    print("ntSynt cases:", len(data["ntSynt"]["cases"]), ends, "| ntsynt_run.py cases:", len(data["ntsynt_run"]))


if __name__ == "__main__":
    main()
