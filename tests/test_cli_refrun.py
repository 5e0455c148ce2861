"""The two Python command lines of the path against the reference's OWN: tests/golden/cli_cases.json holds what bin/ntSynt's main()
and bin/ntsynt_run.py's parse_arguments() did with 166 + 12 argument lists in the build container (tests/golden/make_golden_cli.py:
the reference's code run, `snakemake` and subprocess.call stood in for).  ntsynt_amd/cli.py must end every call the same way -- the
same argparse exit status and message, the same exception, or the same parameters handed on (there: `snakemake --config key=value`,
here: pipeline.run's arguments) after printing the same "Parameter settings" lines; ntsynt_amd/stage_cli.py's stage-3 parser must build
the same namespace."""
import contextlib
import io
import json
import os
import subprocess

import pytest

from ntsynt_amd import cli, stage_cli


@pytest.fixture(scope="module")
def vectors(golden_dir):
    with open(os.path.join(golden_dir, "cli_cases.json")) as fh:
        return json.load(fh)


def _error_line(stderr):
    for line in reversed(stderr.splitlines()):
        if ": error: " in line:
            return line.split(": error: ", 1)[1]
    return None


def _product(argv, monkeypatch):
    handed = {}
    monkeypatch.setattr(cli, "_run", lambda pipeline, fastas, args, device, quiet: handed.update(fastas=fastas, args=args))
    so, se = io.StringIO(), io.StringIO()
    out = {}
    with contextlib.redirect_stdout(so), contextlib.redirect_stderr(se):
        try:
            cli.main(list(argv))
            out["end"] = "ran"
        except SystemExit as exc:
            out.update(end="exit", status=exc.code, error=_error_line(se.getvalue()))
        except subprocess.SubprocessError as exc:
            out.update(end="stage failed", exception=str(exc), stderr=se.getvalue())
        except Exception as exc:                                 # noqa: BLE001
            out.update(end="raised", exception=f"{type(exc).__name__}: {exc}")
    lines = so.getvalue().splitlines()
    if "Parameter settings:" in lines:
        at = lines.index("Parameter settings:")
        out["settings"] = [ln for ln in lines[at + 1:at + 12] if ln.startswith("\t")]
    out["handed"] = handed
    return out


def test_ntsynt_command_line_ends_like_the_reference(vectors, monkeypatch, in_tmp_cwd):
    for name, text in vectors["ntSynt"]["files"].items():
        with open(name, "w") as fh:
            fh.write(text)
    seen = {"exit": 0, "raised": 0, "handed on": 0, "dry run": 0, "duplicate rounds": 0}
    for case in vectors["ntSynt"]["cases"]:
        got = _product(case["argv"], monkeypatch)
        what = f"{case['argv']} ({case['note']})"
        if case["end"] == "exit":
            assert (got["end"], got.get("status"), got.get("error")) == ("exit", case["status"], case["error"]), what
            seen["exit"] += 1
            continue
        if case["end"] == "raised":
            assert (got["end"], got.get("exception")) == ("raised", case["exception"]), what
            seen["raised"] += 1
            continue
        assert got.get("settings") == case["settings"], what
        cmd = case["command"]
        cfg = cmd["config"]
        rounds = cfg["w_rounds"].split()
        if cmd["dry_run"]:
            # -n: the reference hands the plan to `snakemake -n`; this build lists its stages and stops (nothing is handed on)
            assert got["end"] == "ran" and not got["handed"], what
            seen["dry run"] += 1
            continue
        if len(rounds) != len(set(rounds)):
            # the reference's driver lets them through and its stage 3 stops (bin/ntsynt_synteny.py:597-599: the message, exit 1, then
            # bin/ntSynt:166-167's SubprocessError); here the same message and error come before any stage starts
            assert got["end"] == "stage failed" and got["exception"] == "ntSynt failed - check the logs for the error.", what
            assert "Error: duplicate values found in w_rounds!" in got["stderr"], what
            seen["duplicate rounds"] += 1
            continue
        assert got["end"] == "ran" and got["handed"], what
        a = got["handed"]["args"]
        mine = {"references": "[" + ", ".join(got["handed"]["fastas"]) + "]", "kmer": str(a.k), "window": str(a.w), "threads": str(a.t), "fpr": str(a.fpr),
                "prefix": a.prefix, "w_rounds": " ".join(map(str, a.w_rounds)), "indel_merge": str(a.indel), "collinear_merge": str(a.merge),
                "block_size": str(a.block_size), "common": str(not a.no_common), "simplify_graph": str(not a.no_simplify_graph),
                "benchmark": str(a.benchmark), "dev": str(a.dev)}
        assert mine == cfg, what
        assert cmd["cores"] == str(a.t) and cmd["force"] == a.force
        seen["handed on"] += 1
    assert all(n > 0 for n in seen.values()), seen
    assert seen["handed on"] > 80


def test_stage3_parser_builds_the_reference_namespace(vectors):
    parsed = 0
    for case in vectors["ntsynt_run"]:
        se = io.StringIO()
        try:
            with contextlib.redirect_stderr(se):
                ns = vars(stage_cli.run_parser().parse_args(list(case["argv"])))
        except SystemExit as exc:
            assert case["end"] == "exit" and exc.code == case["status"] and _error_line(se.getvalue()) == case["error"], case["argv"]
            continue
        assert case["end"] == "parsed", case["argv"]
        assert {key: ns[key] for key in case["namespace"]} == case["namespace"], case["argv"]
        assert set(ns) - set(case["namespace"]) == {"initial_only", "device"}      # this build's own two
        parsed += 1
    assert parsed >= 6
