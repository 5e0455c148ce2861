#!/usr/bin/env python3
"""The command lines the reference's Snakefile issues -- made by EXECUTING bin/ntsynt_run_pipeline.smk's own Python and expanding its own
`shell:` templates in the build container (/root/reference), over a stand-in for the three things Snakemake supplies:

  config      the `--config key=value` pairs bin/ntSynt hands over (recorded by make_golden_cli.py), each value parsed as YAML -- what
              Snakemake does with command-line config, and why bin/ntSynt writes references='[a.fa, b.fa]' and common=True
  expand()    snakemake.io.expand: the product of the given values, `{{x}}` left as `{x}`
  rules       a `rule name:` block is read as its attributes (input / output / params / threads / resources / shell), each attribute's text
              evaluated as the arguments of a call -- keywords by name, a lone positional as the whole value; callables (the `lambda wildcards:`
              inputs) are called with the rule's wildcards; in a shell template `{params.x}`, `{input.x}`, `{output}`, `{threads}` are filled
              in, lists joined by blanks (an empty list: nothing), as Snakemake formats them

The script half of the file (its first 43 lines: defaults, benchmarking prefix, dict_references) is executed as it stands.  What comes out, per
configuration and per rule instance (one per input genome where the rule has a wildcard): the shell line, its output file and its threads.
Stored: configurations in, command lines out (tests/golden/smk_commands.json) -- no reference source text.  tests/test_smk_refrun.py feeds
these lines to ntsynt_amd/stage_cli.py's parsers (every line must be accepted and mean what the Snakefile's parameters say) and
tests/test_gpu_stages.py runs them against this build's executables.

  python tests/golden/make_golden_smk.py"""
import itertools
import json
import os
import re
import shutil
import sys
import types

import yaml

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
SMK = os.path.join(REF, "bin", "ntsynt_run_pipeline.smk")
SCRIPT_PATH = "<bin>"            # workflow.basedir: where the executables sit


def expand(pattern, **kw):
    "snakemake.io.expand for the forms the file uses"
    patterns = [pattern] if isinstance(pattern, str) else list(pattern)
    keys = list(kw)
    values = [v if isinstance(v, (list, tuple)) or hasattr(v, "__iter__") and not isinstance(v, str) else [v] for v in kw.values()]
    out = []
    for pat in patterns:
        for combo in itertools.product(*[list(v) for v in values]):
            text = pat.replace("{{", "\0").replace("}}", "\1")
            text = text.format(**dict(zip(keys, combo)))
            out.append(text.replace("\0", "{").replace("\1", "}"))
    return out


class Bag(dict):
    "positional items and named items of an input / output / params list, as Snakemake's Namedlist"

    def __init__(self, args, kwargs):
        super().__init__(kwargs)
        self.args = list(args)

    def flat(self):
        out = []
        for v in self.args + list(self.values()):
            out += v if isinstance(v, (list, tuple)) else [v]
        return out


def collect(*args, **kwargs):
    return Bag(args, kwargs)


def split_rules(text):
    "the script half, and {rule name: {attribute: source text}}"
    lines = text.splitlines()
    first = next(i for i, ln in enumerate(lines) if ln.startswith("rule "))
    head = "\n".join(lines[:first])
    rules, name, attr = {}, None, None
    for ln in lines[first:]:
        if not ln.strip() or ln.lstrip().startswith("#"):
            continue
        m = re.match(r"^rule (\w+):\s*$", ln)
        if m:
            name, attr = m.group(1), None
            rules[name] = {}
            continue
        m = re.match(r"^    (\w+):\s*(.*)$", ln)
        if m and not ln.startswith("     "):
            attr = m.group(1)
            rules[name][attr] = m.group(2)
        else:
            rules[name][attr] += "\n" + ln.strip()
    return head, rules


def fmt(value):
    if isinstance(value, (list, tuple)):
        return " ".join(str(v) for v in value)
    return str(value)


def fill(template, input_, output, params, threads):
    def sub(m):
        what = m.group(1)
        if what == "threads":
            return str(threads)
        if what == "output":
            return fmt(output.flat())
        if what == "input":
            return fmt(input_.flat())
        kind, key = what.split(".")
        return fmt({"params": params, "input": input_, "output": output}[kind][key])
    return re.sub(r"\{([\w.]+)\}", sub, template)


def commands(config):
    with open(SMK) as fh:
        head, rules = split_rules(fh.read())
    env = {"config": dict(config), "workflow": types.SimpleNamespace(basedir=SCRIPT_PATH), "expand": expand, "shutil": shutil}
    exec(compile(head, SMK, "exec"), env)                      # noqa: S102 -- the reference's own defaults and benchmarking logic, run as written
    env["collect"] = collect
    refs = list(env["config"]["dict_references"])
    out = []
    for name, attrs in rules.items():
        if "shell" not in attrs:
            continue
        ev = lambda key: eval("collect(" + attrs[key] + ")", env) if key in attrs else Bag([], {})     # noqa: E731, S307
        output = ev("output")
        pattern = fmt(output.flat())
        wild = re.findall(r"\{(\w+)\}", pattern)
        instances = [dict(zip(wild, [r])) for r in refs] if wild else [{}]
        for wc in instances:
            ns = types.SimpleNamespace(**wc)
            call = lambda v: v(ns) if callable(v) else v                                                # noqa: E731
            inp = ev("input")
            inp = Bag([call(v) for v in inp.args], {k: call(v) for k, v in inp.items()})
            par = ev("params")
            res = lambda v: [x.format(**wc) for x in v] if isinstance(v, list) else (v.format(**wc) if isinstance(v, str) else v)   # noqa: E731
            par = Bag([], {k: res(v) for k, v in par.items()})
            outp = Bag([res(v) for v in output.args], {k: res(v) for k, v in output.items()})
            threads = eval(attrs.get("threads", "1"), env)                                              # noqa: S307
            shell = eval(attrs["shell"], env)                                                           # noqa: S307 -- a string literal (with continuations)
            line = fill(shell, inp, outp, par, threads)
            out.append({"rule": name, "wildcards": wc, "threads": threads, "output": outp.flat(), "shell": " ".join(line.split())})
    return out


def main():
    with open(os.path.join(OUT, "cli_cases.json")) as fh:
        cli = json.load(fh)["ntSynt"]["cases"]
    picked, seen = [], set()
    for case in cli:
        if case.get("end") != "ran":
            continue
        cfg = case["command"]["config"]
        key = (cfg["common"], cfg["simplify_graph"], cfg["dev"], cfg["benchmark"], cfg["w_rounds"], cfg["collinear_merge"], cfg["references"])
        if key in seen or len(picked) >= 14:
            continue
        seen.add(key)
        picked.append(case)
    runs = []
    for case in picked:
        raw = case["command"]["config"]
        config = {k: yaml.safe_load(v) for k, v in raw.items()}
        if config.get("benchmark"):
            continue                                            # (memusg / time lookups depend on the box: not recorded)
        runs.append({"ntSynt_argv": case["argv"], "config": raw, "commands": commands(config)})
    # the experimental repeat filter is not reachable from bin/ntSynt: one configuration by hand (smk:17,65-85)
    raw = dict(picked[0]["command"]["config"], repeat="True")
    runs.append({"ntSynt_argv": None, "config": raw, "commands": commands({k: yaml.safe_load(v) for k, v in raw.items()})})
    with open(os.path.join(OUT, "smk_commands.json"), "w") as fh:
        json.dump({"script_path": SCRIPT_PATH, "runs": runs}, fh, indent=1, sort_keys=True)
        fh.write("\n")
    for r in runs[:2] + runs[-1:]:
        print(r["config"])
        for c in r["commands"]:
            print("   ", c["rule"], c["wildcards"], "|", c["shell"])
    print(len(runs), "configurations,", sum(len(r["commands"]) for r in runs), "command lines")


if __name__ == "__main__":
    main()
