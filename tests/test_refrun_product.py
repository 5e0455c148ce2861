"""The product's graph-stage rules (ntsynt_amd/synteny.py: the host-array engine, twin of the device engine and the code the device
engine inherits its per-block passes from) against the reference's OWN code -- tests/golden/refrun/, recorded runs of
bin/ntsynt_synteny.py's main_synteny (tests/golden/make_golden_refrun.py; stand-ins and their assumptions listed there).

No GPU: the engine's two device-side inputs come from test doubles, as in tests/test_engine_cpu.py (the graph build from
tests/graph_ref.py, the masked re-sketch from the oracle's indexlr restatement), so what is compared with the reference's run is the
rule set itself, step by step (tests/refrun.py: HostLockstep): bubble removal, weight filter and flagged pairs, erosion, paths,
orientation vote, indel split, the four-minimizer rule, block extents, mask intervals, the refinement round's list filter and cuts,
the position table, and finally the bytes of the TSVs, the interarrival file and the --dev warnings.  tests/test_gpu_refrun.py runs the
same scenarios through the HIP path."""
import contextlib
import io
import json
import os

import numpy as np
import pytest

from ntsynt_amd.graph import edge_degrees, walk_paths
from ntsynt_amd.synteny import SyntenyEngine
from oracle import nts_oracle as O
from tests import refrun
from tests.graph_ref import build_graph_numpy
from tests.helpers import oracle_flat
from tests.refrun import HostLockstep, Scenario, drive_host


def host_engine(sc, fastas, native=True):
    m = sc.meta
    k, w = m["k"], m["w"]
    genomes = [O.read_fasta(p) for p in fastas]
    bf = O.common_bf({p: g for p, g in zip(fastas, genomes)}, k, 0.025) if m.get("common", True) else None
    tsvs = [f"{os.path.basename(p)}.k{k}.w{w}.tsv" for p in fastas]
    # stage 3's repeat filter (ntsynt_run.py --filter, S:172-185, S:601-607) the way ntsynt_amd.pipeline hands it to the engine: `Indexlr`
    # = the refinement sketches are made with it as filter-out; `Filter` = every list, initial and refined, is screened before the engine
    # sees it (nts_mx_screen on the device; here the same rule on the test doubles' lists)
    mode = getattr(sc, "filter_mode", None)
    rep = sc.repeat_filter(genomes) if mode else None

    def screened(g, flat):
        if mode != "Filter":
            return flat
        h1, rec, pos = flat
        keep = np.array([not O.bf_contains(rep, O.hash_kmer(bytes(g.record(int(r))[int(p):int(p) + k]))[0]) for r, p in zip(rec, pos)], dtype=bool)
        return h1[keep], rec[keep], pos[keep]

    initial = [screened(g, oracle_flat(O.minimize(g, k, w, bf))) for g in genomes]

    def sketch_fn(i, masks, new_w):
        g = genomes[i]
        seqs = []
        for r in range(len(g.names)):
            buf = bytearray(g.record(r))
            for mr, s, e in masks:
                if mr == r:
                    s, e = max(0, s), min(len(buf), e)
                    if e > s:
                        buf[s:e] = b"N" * (e - s)
            seqs.append(bytes(buf))
        masked = O.Genome(g.names, seqs)
        return screened(masked, oracle_flat(O.minimize(masked, k, new_w, bf, repeat=rep if mode == "Indexlr" else None)))

    eng = SyntenyEngine(tsvs, [g.names for g in genomes], k, w, m["w_rounds"], m["indel"], m["merge"], m["z"], sc.prefix,
                        build_graph_numpy, sketch_fn, walk_paths, degree_fn=edge_degrees if native else None, n=sc.min_weight,
                        dev=True, interarrivals=True, simplify=getattr(sc, "simplify", True), m=getattr(sc, "m", 90))
    return eng, initial


@pytest.mark.parametrize("name", refrun.scenario_names())
def test_host_engine_in_lockstep_with_the_reference_run(name, in_tmp_cwd):
    sc = Scenario(name)
    fastas = sc.unpack(str(in_tmp_cwd))
    eng, initial = host_engine(sc, fastas)
    lock = HostLockstep(eng, sc)
    err = io.StringIO()
    with contextlib.redirect_stderr(err), sc.ends_like_the_reference():
        drive_host(lock, initial)
    out = eng.outputs
    assert out[f"{sc.prefix}.pre-collinear-merge.synteny_blocks.tsv"] == sc.expected("pre-collinear-merge.synteny_blocks.tsv")
    assert out[f"{sc.prefix}.synteny_blocks.tsv"] == sc.expected("synteny_blocks.tsv")
    # the interarrival file: same lines; the block order follows ntJoin's component order, which nothing pins (synteny.py:647-651)
    assert sorted(out[f"{sc.prefix}.interarrivals.tsv"].splitlines()) == sorted(sc.expected("interarrivals.tsv").splitlines())
    assert [ln for ln in err.getvalue().splitlines() if ln.startswith("WARNING")] == sc.meta["warnings"]
    assert lock.checked["paths"] > (0 if sc.stopped else 10) and lock.checked["filtered_lists"] > 0 and lock.checked["valid_minimizers"] > 0


@pytest.mark.parametrize("name", refrun.scenario_names())
def test_host_engine_run_writes_the_reference_runs_bytes(name, in_tmp_cwd):
    "SyntenyEngine.run itself (the lockstep test above drives its steps one by one)"
    sc = Scenario(name)
    fastas = sc.unpack(str(in_tmp_cwd))
    eng, initial = host_engine(sc, fastas)
    err = io.StringIO()
    with contextlib.redirect_stderr(err), sc.ends_like_the_reference():
        eng.run(initial)
    out = eng.outputs
    assert out[f"{sc.prefix}.pre-collinear-merge.synteny_blocks.tsv"] == sc.expected("pre-collinear-merge.synteny_blocks.tsv")
    assert out[f"{sc.prefix}.synteny_blocks.tsv"] == sc.expected("synteny_blocks.tsv")
    assert [ln for ln in err.getvalue().splitlines() if ln.startswith("WARNING")] == sc.meta["warnings"]


# ------------------------------------------------------------------------------------------------ per-function vectors
@pytest.fixture(scope="module")
def unit(golden_dir):
    with open(os.path.join(golden_dir, "unit_cases.json")) as fh:
        return json.load(fh)


def test_find_fa_name_through_the_stage_executable(unit):
    "S:108-116 (the TSV name without its .k<k>.w<w>.tsv tail names the FASTA; anything else ends the run) == stage_cli.fasta_name_of"
    from ntsynt_amd.stage_cli import fasta_name_of
    for name, want in unit["find_fa_name"]:
        if isinstance(want, dict):
            with pytest.raises(SystemExit) as e, contextlib.redirect_stdout(io.StringIO()):
                fasta_name_of(name)
            assert e.value.code == want["exit"]
        else:
            assert fasta_name_of(name) == want


def test_block_extents_and_interiors(unit):
    "A:17-23 (start, end) and S:194-203 (the interior handed to the interval index) as the engine computes them from a block's two ends"
    eng = object.__new__(SyntenyEngine)
    from ntsynt_amd.synteny import Block
    for c in unit["assembly_block"]:
        eng.k, eng.G = c["k"], 1
        b = Block(np.zeros(0, np.int64), [0], ["+"], None, [c["mx"][0][1]], [c["mx"][-1][1]], len(c["mx"]))
        assert (eng._start(b, 0), eng._end(b, 0)) == (c["start"], c["end"])
        assert eng._end(b, 0) - eng._start(b, 0) == c["length"]
    from ntsynt_amd.synteny_device import DeviceSyntenyEngine
    dev = object.__new__(DeviceSyntenyEngine)
    dev.G = 1
    for c in unit["update_intervals"]:
        tb = {"first_pos": np.array([[c["p1"]]], np.int64), "last_pos": np.array([[c["p2"]]], np.int64), "rec": np.array([[0]], np.uint32)}
        comp_s, comp_mx = dev._spans(tb)[0]
        got = [[int(s), int(e), 1] for s, e in zip(comp_s.tolist(), comp_mx.tolist())]
        assert ([[1, 2, 1]] if c["pre"] else []) + got == c["out"]
