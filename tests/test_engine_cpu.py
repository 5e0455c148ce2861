"""Host logic of the HIP path (ntsynt_amd/synteny.py: rows C3-C12) against the oracle, without a GPU.

The engine gets its two device-side inputs from test doubles here -- the graph build from
tests/graph_ref.py (numpy) and the masked re-sketch from the oracle's indexlr restatement -- so that
what is compared is exactly the host-side rule set: simplification order, weight filter, path
orientation, block splitting, refinement bookkeeping, erosion, collinear merge, TSV bytes."""
import os

import numpy as np
import pytest

from ntsynt_amd import synth
from ntsynt_amd.graph import walk_chains
from ntsynt_amd.synteny import SyntenyEngine
from oracle import nts_oracle as O
from oracle import synteny_oracle as SO
from tests.graph_ref import build_graph_numpy
from tests.helpers import oracle_flat


def run_both(tmp_path, paths, k, w, w_rounds, indel, merge, block_size, simplify=True, common=True):
    os.makedirs(tmp_path / "ora", exist_ok=True)
    os.makedirs(tmp_path / "eng", exist_ok=True)
    os.chdir(tmp_path / "ora")
    ora = SO.run_pipeline(paths, k=k, w=w, w_rounds=w_rounds, indel=indel, merge=merge, block_size=block_size,
                          prefix="p", simplify=simplify, common=common)
    os.chdir(tmp_path / "eng")
    genomes = [O.read_fasta(p) for p in paths]
    bf = ora.bf
    tsvs = [f"{os.path.basename(p)}.k{k}.w{w}.tsv" for p in paths]
    initial = [oracle_flat(O.minimize(g, k, w, bf)) for g in genomes]

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
        return oracle_flat(O.minimize(O.Genome(g.names, seqs), k, new_w, bf))

    eng = SyntenyEngine(tsvs, [g.names for g in genomes], k, w, w_rounds, indel, merge, block_size, "p",
                        build_graph_numpy, sketch_fn, walk_chains, simplify=simplify)
    out = eng.run(initial)
    return ora.outputs, out


CASES = [
    # (n_genomes, total_bp, contigs, divergence, seed, k, w, w_rounds, indel, merge, block)
    (2, 600_000, 2, 0.01, 1, 24, 1000, [100, 10], 500, 3000, 500),
    (3, 900_000, 3, 0.01, 2, 24, 1000, [100, 10], 500, 3000, 500),
    (3, 900_000, 2, 0.02, 3, 20, 500, [100], 10000, "3w", 500),
    (4, 700_000, 2, 0.005, 4, 24, 400, [100, 10], 500, 1000, 300),
    (2, 1_200_000, 1, 0.03, 5, 24, 1000, [250, 100], 50000, 100000, 1000),
    (3, 600_000, 2, 0.01, 6, 32, 250, [50, 5], 200, "2w", 200),
    (2, 1_500_000, 2, 0.005, 9, 24, 200, [50, 10], 5000, 20000, 300),
    (3, 1_200_000, 2, 0.005, 12, 24, 150, [40, 10], 3000, "40w", 200),
    (2, 1_000_000, 1, 0.002, 15, 20, 100, [20, 5], 2000, 10000, 100),
]


@pytest.mark.parametrize("case", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_engine_matches_oracle(tmp_path, case):
    n, bp, ctg, div, seed, k, w, rounds, indel, merge, block = case
    cwd = os.getcwd()
    try:
        paths = synth.make_family(str(tmp_path), n, bp, ctg, div, seed=seed, n_runs=(seed % 2 == 0),
                                  micro=(12 if seed % 3 == 0 else 0))
        exp, got = run_both(tmp_path, paths, k, w, rounds, indel, merge, block)
    finally:
        os.chdir(cwd)
    assert set(got) == {"p.synteny_blocks.tsv", "p.pre-collinear-merge.synteny_blocks.tsv"}
    for name in got:
        assert got[name] == exp[name], name
    assert len(got["p.synteny_blocks.tsv"].splitlines()) >= 2 * n


def test_engine_without_simplification(tmp_path):
    cwd = os.getcwd()
    try:
        paths = synth.make_family(str(tmp_path), 3, 600_000, 2, 0.02, seed=11)
        exp, got = run_both(tmp_path, paths, 24, 500, [100, 10], 500, 3000, 500, simplify=False)
    finally:
        os.chdir(cwd)
    for name in got:
        assert got[name] == exp[name], name


def test_walk_chains_shapes():
    # two paths, one cycle, one branching component, one isolated vertex
    eu = np.array([0, 1, 3, 5, 6, 7, 8, 9, 10], dtype=np.int64)
    ev = np.array([1, 2, 4, 6, 7, 5, 9, 10, 11], dtype=np.int64)
    eu = np.concatenate((eu, [9]))
    ev = np.concatenate((ev, [12]))       # vertex 9 gets degree 3
    off, verts = walk_chains(14, eu, ev)
    paths = [verts[off[i]:off[i + 1]].tolist() for i in range(off.size - 1)]
    assert paths == [[0, 1, 2], [3, 4]]
