"""The reference-generated vectors (tests/golden/make_golden.py imported the reference's own Python: sort -> merge_collinear_blocks ->
z filter -> merge_collinear_blocks -> get_block_string, /root/reference/bin/ntsynt_synteny.py:428-472, bin/synteny_block.py:48-109;
the demo's pre-merge and final TSVs are the reference's own test fixtures) asserted against the PRODUCT's code directly -- not through
product == oracle:

  * the host-array engine (ntsynt_amd/synteny.py: _sorted, _merge, _long_enough, _text) -- rows C10, C12-merge;
  * the device engine's host-side passes over the block table (ntsynt_amd/synteny_device.py: _sorted, _merge = nts_blocks_merge,
    _long_mask, _emit = nts_blocks_text in libntsynt_hip.so) -- the code `bin/ntSynt` runs;
  * the orientation vote (C7: nts_path_scan's rising steps + SyntenyEngine._orient_codes) and the indel spread (C8: nts_path_scan's
    `over` flags) on block_cases.json.

No GPU: these are host passes (the library loads without a device; nothing here creates a context)."""
import collections
import json
import os

import numpy as np
import pytest

from ntsynt_amd import _lib
from ntsynt_amd.graph import scan_paths
from ntsynt_amd.synteny import Block, SyntenyEngine
from ntsynt_amd.synteny_device import DeviceSyntenyEngine


class _HostOnly:
    "what the block-table passes need of a context: the library (nts_blocks_merge / nts_blocks_text are host functions)"

    def __init__(self):
        self.lib = _lib.load()


def _engines(files, contigs, k, bp, cm, z):
    "the product's two engines on `files` (minimizer-TSV names) with contigs[file] = record names; no graph, no device"
    names = [contigs[f] for f in files]
    host = SyntenyEngine(files, names, k, 1000, [100, 10], bp, cm, z, "x", None, None, None, scan_fn=scan_paths)
    dev = object.__new__(DeviceSyntenyEngine)                  # (its constructor would create the device graph)
    SyntenyEngine.__init__(dev, files, names, k, 1000, [100, 10], bp, cm, z, "x", None, None, None, scan_fn=False)
    dev.ctx, dev._ctg_rank, dev._blob = _HostOnly(), None, None
    return host, dev


def _from_rows(blocks):
    "golden blocks [{asm: {file: (contig, ori, first_pos, last_pos, n)}, reason}] -> (files, contigs)"
    files = sorted({f for b in blocks for f in b["asm"]})
    contigs = {f: sorted({b["asm"][f][0] for b in blocks}) for f in files}
    return files, contigs


def _host_blocks(eng, blocks):
    out = []
    for b in blocks:
        rec, ori, fp, lp, n = [], [], [], [], 0
        for f in eng.files:                                     # engine order (descending file names, S:34)
            ctg, o, first, last, n = b["asm"][f]
            rec.append(eng.contigs[eng.files.index(f)].index(ctg))
            ori.append(o)
            fp.append(first)
            lp.append(last)
        out.append(Block(np.zeros(0, np.int64), rec, ori, b["reason"], fp, lp, n))
    return out


REASONS = {None: 0, "id_change": 1, "ori_change": 2, "inconsistent_order": 3, "indel": 4, "merge": 5}


def _device_table(eng, blocks):
    n = len(blocks)
    # dtypes as DeviceSyntenyEngine._blocks hands the table on (include/ntsynt_hip.h: nts_blocks_merge / nts_blocks_text read them raw)
    tb = {"n": n, "rec": np.zeros((eng.G, n), np.uint32), "first_pos": np.zeros((eng.G, n), np.int64), "last_pos": np.zeros((eng.G, n), np.int64),
          "ori": np.zeros((eng.G, n), np.uint8), "n_mx": np.zeros(n, np.int64), "reason": np.zeros(n, np.uint8)}
    for i, b in enumerate(blocks):
        for a, f in enumerate(eng.files):
            ctg, o, first, last, cnt = b["asm"][f]
            tb["rec"][a, i] = eng.contigs[a].index(ctg)
            tb["ori"][a, i] = "+-?".index(o)
            tb["first_pos"][a, i], tb["last_pos"][a, i] = first, last
            tb["n_mx"][i] = cnt
        tb["reason"][i] = REASONS[b["reason"]]
    return tb


def _host_texts(eng, blocks):
    ordered = eng._sorted(_host_blocks(eng, blocks))
    pre = "".join(eng._text(b, i, False) for i, b in enumerate(ordered))
    merged = eng._merge(ordered)
    merged = [b for b in merged if eng._long_enough(b)]
    if merged:
        merged = eng._merge(merged)
    eng._emit("final.tsv", merged, verbose=True)
    return pre, eng.outputs["final.tsv"]


def _device_texts(eng, blocks):
    ordered = eng._sorted(_device_table(eng, blocks))
    z = eng.z
    eng.z = -(1 << 40)                                          # the golden pre-merge text lists every block (run_merge in make_golden.py)
    eng._emit("pre.tsv", ordered)
    eng.z = z
    merged = eng._merge(ordered)
    merged = eng._take(merged, np.flatnonzero(eng._long_mask(merged)))
    if merged["n"]:
        merged = eng._merge(merged)
    eng._emit("final.tsv", merged, verbose=True)
    return eng.outputs["pre.tsv"], eng.outputs["final.tsv"]


def _case_blocks(case):
    out = []
    for b in case["blocks"]:
        asm = {}
        for f, d in b["asm"].items():
            mx = d["mx"]
            asm[f] = (d["contig"], d["ori"], mx[0][1], mx[-1][1], len(mx))
        out.append({"asm": asm, "reason": b["reason"]})
    return out


def test_merge_random_goldens_through_both_product_engines(golden_dir, in_tmp_cwd):
    "40 random block lists: the reference's sorted order, pre-merge text, merge decisions, reasons and final text"
    cases = json.load(open(os.path.join(golden_dir, "merge_cases.json")))
    assert len(cases) >= 40
    n_merged = 0
    for c in cases:
        blocks = _case_blocks(c)
        files, contigs = _from_rows(blocks)
        host, dev = _engines(files, contigs, c["k"], c["bp"], c["collinear_merge"], c["z"])
        pre, out = _host_texts(host, blocks)
        assert pre == c["pre_text"], ("host engine, pre-merge text", c["name"])
        assert out == c["final_text"], ("host engine, final text", c["name"])
        pre, out = _device_texts(dev, blocks)
        assert pre == c["pre_text"], ("nts_blocks_text, pre-merge text", c["name"])
        assert out == c["final_text"], ("nts_blocks_merge / nts_blocks_text, final text", c["name"])
        n_merged += host.stats["merged"] > 0 and dev.stats["merged"] == host.stats["merged"]
    assert n_merged > 10                                        # the vectors do exercise merging, and both engines count the same merges


@pytest.mark.parametrize("stem,k", [("celegans-A-ntSynt", 24), ("celegans-A-B-ntSynt", 20)])
def test_demo_pre_merge_tsv_to_final_tsv_through_both_product_engines(golden_dir, in_tmp_cwd, stem, k):
    "the reference demo's own fixtures (tests/ntsynt_tests.py:40-59): pre-collinear-merge TSV -> final TSV, byte for byte"
    rows = collections.OrderedDict()
    for line in open(os.path.join(golden_dir, stem + ".pre-collinear-merge.synteny_blocks.tsv")):
        num, asm, ctg, start, end, ori, n = line.rstrip("\n").split("\t")
        first, last = (int(start), int(end) - k) if ori == "+" else (int(end) - k, int(start))
        rows.setdefault(int(num), {})[asm + ".k1.w1.tsv"] = (ctg, ori, first, last, int(n))
    blocks = [{"asm": asm, "reason": None} for asm in rows.values()]
    files, contigs = _from_rows(blocks)
    want_pre = open(os.path.join(golden_dir, stem + ".pre-collinear-merge.synteny_blocks.tsv")).read()
    want = open(os.path.join(golden_dir, stem + ".synteny_blocks.tsv")).read()
    host, dev = _engines(files, contigs, k, 500, 3000, 500)
    assert _host_texts(host, blocks) == (want_pre, want)
    assert _device_texts(dev, blocks) == (want_pre, want)


def test_orientation_vote_and_indel_spread_through_the_product_scan(golden_dir):
    """block_cases.json: 200 position lists with the reference's orientation call at their threshold m (synteny_block.py:48-65) through
    nts_path_scan's rising-step count + SyntenyEngine._orient_codes; 100 pairs of minimizers in five assemblies with the reference's
    max_difference (ntsynt_synteny.py:365-389) through nts_path_scan's indel flag at bp = spread and bp = spread - 1 (S:399: `>`)"""
    d = json.load(open(os.path.join(golden_dir, "block_cases.json")))
    seen = collections.Counter()
    for c in d["orientation"]:
        pos = np.array(c["pos"], np.int64)[None, :]
        start, n_up, over = scan_paths(np.zeros_like(pos), pos, np.array([0, pos.shape[1]], np.uint64), np.arange(pos.shape[1]), 1 << 60)
        eng = object.__new__(SyntenyEngine)
        eng.m = c["m"]
        code = eng._orient_codes(n_up[0], np.array([pos.shape[1] - 1]))
        assert "+-?"[int(code[0])] == c["ori"], c
        assert start[0] == 0 and not over.any()
        seen[c["ori"]] += 1
    assert min(seen["+"], seen["-"]) > 10 and seen["?"] > 0     # the vectors hold all three outcomes
    for c in d["max_difference"]:
        v_pos = np.array([c["p1"], c["p2"]], np.int64).T.copy()           # [assembly][vertex]
        off, verts = np.array([0, 2], np.uint64), np.arange(2)
        for bp, cut in ((c["spread"], False), (c["spread"] - 1, True)):
            _, _, over = scan_paths(np.zeros_like(v_pos), v_pos, off, verts, bp)
            assert bool(over[0]) is cut, (c, bp)
