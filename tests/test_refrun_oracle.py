"""The CPU oracle's graph stage (oracle/synteny_oracle.py) against the reference's OWN code.

tests/golden/refrun/ holds runs of bin/ntsynt_synteny.py's NtSyntSynteny.main_synteny (with synteny_block.py / assembly_block.py),
imported unmodified in the build container and executed over stand-ins for the third-party modules the image lacks
(tests/golden/make_golden_refrun.py lists each stand-in's assumed semantics: ntJoin's helpers = this oracle's restatement,
igraph, ncls, intervaltree, bedtools slop/maskfasta, seqtk).  Asserted here:

  * end to end: the oracle pipeline on the scenario's FASTA files writes the reference run's pre-collinear-merge and final TSVs, its
    interarrival file and its --dev overlap warnings, byte for byte;
  * in lockstep: every call the reference made to run_graph_simplification, find_synteny_blocks, check_for_indels,
    filter_synteny_blocks, get_synteny_bed_lists (+ the mask intervals), generate_new_minimizers, find_mx_in_blocks,
    filter_minimizers_synteny_blocks, update_list_mx_info, filter_graph_global_flag_overlaps, refine_graph and
    merge_collinear_blocks, against the oracle method that restates it, in call order;
  * the small per-function vectors of tests/golden/unit_cases.json (AssemblyBlock accessors, SyntenyBlock's walk, update_intervals,
    filter_minimizers_synteny_blocks, update_list_mx_info)."""
import contextlib
import io
import json
import os
from collections import defaultdict

import pytest

from oracle import nts_oracle as O
from oracle import synteny_oracle as SO
from tests import refrun
from tests.refrun import Cursor, Scenario, edge_digest, lists_digest, undelta


def _digest(g):
    return edge_digest((e[0], e[1], e[2]) for e in g.edges)


def _edge_set(g):
    return {frozenset((e[0], e[1])) for e in g.edges}


def _blocks(blocks, positions=True):
    out = []
    for b in blocks:
        lists = [[h for h, _ in ab.minimizers] for ab in b.asm.values()]
        assert all(x == lists[0] for x in lists)
        rows = []
        for a, ab in b.asm.items():
            row = (a, ab.contig_id, ab.ori)
            if positions:
                row += (tuple(p for _, p in ab.minimizers),)
            rows.append(row)
        out.append((tuple(lists[0]), tuple(sorted(rows))))
    return out


def _want_blocks(cur, blocks, positions=True):
    out = []
    for b in blocks:
        rows = []
        for a, d in b["asm"].items():
            row = (a, d["contig"], d["ori"])
            if positions:
                row += (tuple(undelta(d["dpos"])),)
            rows.append(row)
        out.append((tuple(cur.names(b["mx"])), tuple(sorted(rows))))
    return out


class TracedOracle(SO.SyntenyOracle):
    "the oracle engine with every restated function held against the reference's recorded call"

    def attach(self, scenario):
        self.cur = Cursor(scenario)
        self.seen = defaultdict(int)
        self.dev = True
        return self

    def simplify_graph(self, graph):                                  # S:566-590
        cur, ev = self.cur, self.cur.take("run_graph_simplification")
        assert _digest(graph) == ev["graph_before"]
        w0 = {frozenset((e[0], e[1])): e[2] for e in graph.edges}
        out = super().simplify_graph(graph)
        assert set(graph.adj) - set(out.adj) == set(cur.names(ev["removed"]))
        assert {frozenset((e[0], e[1])) for e in graph.edges if w0[frozenset((e[0], e[1]))] != e[2]} == cur.pairs(ev["promoted"])
        assert _digest(out) == ev["graph_after"] and _digest(graph) == ev["input_after"]
        self.seen["bubbles"] += len(ev["removed"])
        return out

    def _blocks_of_path(self, path):                                  # S:66-106
        cur, ev = self.cur, self.cur.take("find_synteny_blocks")
        assert list(path) == cur.names(ev["path"])
        before = set(self.graph.adj)
        out = super()._blocks_of_path(path)
        assert _blocks(out, positions=False) == _want_blocks(cur, ev["blocks"], positions=False)
        assert before - set(self.graph.adj) == set(cur.names(ev["removed"]))
        self.seen["unoriented"] += bool(ev["removed"])
        self.seen["shortened"] += bool(ev["blocks"]) and len(ev["blocks"][0]["mx"]) < len(ev["path"])
        return out

    def split_indels(self, blocks):                                   # S:391-409
        cur, ev = self.cur, self.cur.take("check_for_indels")
        assert ev["n_in"] == len(blocks) and ev["bp"] == self.bp
        before = _edge_set(self.graph)
        out = super().split_indels(blocks)
        assert _blocks(out) == _want_blocks(cur, ev["blocks_out"])
        assert before - _edge_set(self.graph) == cur.pairs(ev["removed_edges"])
        self.seen["indel_edges"] += len(ev["removed_edges"])
        self._last_split = out
        return out

    def drop_small(self, blocks, min_mx):                             # S:411-426
        cur, ev = self.cur, self.cur.take("filter_synteny_blocks")
        assert ev["threshold"] == min_mx and ev["n_in"] == len(blocks)
        before = set(self.graph.adj)
        out = super().drop_small(blocks, min_mx)
        assert [id(b) for b in out] == [id(blocks[i]) for i in ev["kept"]]
        assert before - set(self.graph.adj) == set(cur.names(ev["removed"]))
        assert _digest(self.graph) == ev["graph_after"]
        self.seen["small"] += len(blocks) - len(out)
        return out

    def mask_intervals(self, blocks, w):                              # S:118-146
        cur = self.cur
        ev_b = cur.take("get_synteny_bed_lists")
        beds = {}
        for blk in blocks:
            for a, ab in blk.asm.items():
                beds.setdefault(a, {}).setdefault(ab.contig_id, []).append([ab.start(), ab.end()])
        assert beds == ev_b["beds"]                                   # A:17-23 extents, grouped in block order like S:118-132
        ev = cur.take("mask_assemblies_with_synteny_extents")
        assert ev["w"] == w
        out = super().mask_intervals(blocks, w)
        want = {}
        for pf in ev["per_fasta"]:
            tsv = [a for a in self.files if a.startswith(pf["fasta"] + ".k")][0]
            # S:142-143: what the reference's own filter hands to bedtools
            lim = max(2 * w, w + self.k + 1)
            handed = sorted([c, s, e] for c, lst in ev_b["beds"][tsv].items() for s, e in lst if e - s > lim)
            assert sorted(x[:3] for x in pf["handed_to_slop"]) == handed
            if pf["masked"]:
                want[tsv] = sorted(pf["masked"])
        got = {a: sorted([c, s, e] for c, lst in d.items() for s, e in lst) for a, d in out.items()}
        assert got == want
        self.seen["masked"] += sum(len(v) for v in want.values())
        self.seen["emptied_by_slop"] += sum(len(pf["handed_to_slop"]) - len(pf["masked"]) for pf in ev["per_fasta"])
        return out

    def new_minimizers(self, blocks, new_w, prev_w):                  # S:532-541: generate_new_minimizers closes the sketch seam
        self._gen = None
        self._sketched = {}
        return super().new_minimizers(blocks, new_w, prev_w)

    def sketch_masked(self, asm, ctg_masks, new_w):
        out = super().sketch_masked(asm, ctg_masks, new_w)
        self._sketched[asm] = out
        return out

    def block_marks(self, blocks):                                    # S:205-226, S:194-203
        cur = self.cur
        ev_g = cur.take("generate_new_minimizers")
        assert lists_digest({a: lists for a, (info, lists) in self._sketched.items()}, {a: info for a, (info, lists) in self._sketched.items()}) == ev_g["sketch"]
        ev = cur.take("find_mx_in_blocks")
        terminal, internal, spans = super().block_marks(blocks)
        assert terminal == set(cur.names(ev["terminal"])) and internal == set(cur.names(ev["internal"]))
        assert {a: {c: sorted(list(x) for x in v) for c, v in d.items()} for a, d in spans.items()} == ev["intervals"]
        return terminal, internal, spans

    def _filter_lists_checked(self, list_mxs, internal, new_info, spans):
        cur, ev = self.cur, self.cur.take("filter_minimizers_synteny_blocks")
        out = SO.SyntenyOracle.filter_lists(list_mxs, internal, new_info, spans)
        assert lists_digest(out) == ev["lists_out"] and {a: len(v) for a, v in out.items()} == ev["n_lists_out"]
        self.seen["list_cuts"] += sum(len(v) for v in out.values()) - sum(len(v) for v in list_mxs.values())
        return out

    def update_info(self, filt, new_info):                            # S:282-290
        cur, ev = self.cur, self.cur.take("update_list_mx_info")
        assert lists_digest(filt) == ev["lists_common"]
        super().update_info(filt, new_info)
        valid = {h for v in filt.values() for lst in v for h in lst}
        assert len(valid) == ev["n_valid"]
        assert lists_digest({a: [sorted(h for h in valid if h in self.list_mx_info[a])] for a in new_info}, self.list_mx_info) == ev["info_after"]

    def refine_graph(self, flagged):                                  # S:292-303 result + S:343-362
        cur = self.cur
        ev_f = cur.take("filter_graph_global_flag_overlaps")
        assert [list(x) for x in flagged] == [cur.names(x) for x in ev_f["flagged"]]       # in edge order
        assert _digest(self.graph) == ev_f["graph_after"]
        ev = cur.take("refine_graph")
        before = _edge_set(self.graph)
        out = super().refine_graph(flagged)
        assert before - _edge_set(out) == cur.pairs(ev["eroded_edges"])
        assert _digest(out) == ev["graph_after"]
        self.seen["eroded"] += len(ev["eroded_edges"])
        return out

    def merge_collinear(self, blocks):                                # S:428-472
        if not blocks and self.cur.sc.stopped and self.cur.done():   # S:437: the reference's run ended inside this call
            return super().merge_collinear(blocks)
        ev = self.cur.take("merge_collinear_blocks")
        assert ev["n_in"] == len(blocks)
        out = super().merge_collinear(blocks)
        assert [b.broken_reason for b in out] == ev["reasons"]
        self.seen["merged"] += len(blocks) - len(out)
        return out


def _run_traced(sc, tmp, use_repeat=True):
    fastas = sc.unpack(str(tmp))
    m = sc.meta
    k, w = m["k"], m["w"]
    genomes = {p: O.read_fasta(p) for p in fastas}
    bf = O.common_bf(genomes, k, 0.025) if m.get("common", True) else None
    tables, by_tsv = {}, {}
    for p in fastas:
        tsv = f"{os.path.basename(p)}.k{k}.w{w}.tsv"
        O.write_indexlr_tsv(tsv, genomes[p], O.minimize(genomes[p], k, w, bf), k)
        tables[tsv] = SO.read_minimizers_tsv(tsv)
        by_tsv[tsv] = genomes[p]
    eng = TracedOracle(list(tables), by_tsv, k, w, m["w_rounds"], m["indel"], m["merge"], m["z"], sc.prefix, bf=bf, n=sc.min_weight,
                       interarrivals=True, simplify=sc.simplify, m=sc.m).attach(sc)
    # the instance's filter_lists is reached through the class (a staticmethod): route it through the checked version
    eng.filter_lists = eng._filter_lists_checked
    if sc.filter_mode and use_repeat:
        rep = sc.repeat_filter([genomes[p] for p in fastas])
        if sc.filter_mode == "Indexlr":                               # S:172-180: the refinement rounds' indexlr runs with -r
            eng.refine_repeat = rep
        else:                                                        # S:184-185, S:605-607: read_minimizers(file, repeat_bf), every time
            eng.screen_repeat = rep
            tables = {t: SO.read_minimizers_tsv(t, repeat_bf=rep) for t in tables}
    eng.load(tables)
    err = io.StringIO()
    with contextlib.redirect_stderr(err), sc.ends_like_the_reference():
        eng.main()
    return eng, [ln for ln in err.getvalue().splitlines() if ln.startswith("WARNING")]


@pytest.mark.parametrize("name", refrun.scenario_names())
def test_oracle_in_lockstep_with_the_reference_run(name, in_tmp_cwd):
    sc = Scenario(name)
    eng, warnings = _run_traced(sc, in_tmp_cwd)
    assert eng.cur.done(), "calls of the reference's run the oracle never made"
    assert eng.outputs[f"{sc.prefix}.pre-collinear-merge.synteny_blocks.tsv"] == sc.expected("pre-collinear-merge.synteny_blocks.tsv")
    assert eng.outputs[f"{sc.prefix}.synteny_blocks.tsv"] == sc.expected("synteny_blocks.tsv")
    assert eng.outputs[f"{sc.prefix}.interarrivals.tsv"] == sc.expected("interarrivals.tsv")
    assert warnings == sc.meta["warnings"]
    assert eng.seen["masked"] > 0


@pytest.mark.parametrize("name", [n for n in refrun.scenario_names() if Scenario(n).filter_mode])
def test_the_repeat_filter_matters_in_its_scenarios(name, in_tmp_cwd):
    "negative control: the same run without the repeat filter leaves the reference's trace (else these scenarios prove nothing about --filter)"
    with pytest.raises(AssertionError):
        _run_traced(Scenario(name), in_tmp_cwd, use_repeat=False)


def test_the_scenarios_reach_every_rule(in_tmp_cwd):
    "the recorded runs together exercise each branch the restatement has (otherwise the lockstep proves little)"
    seen = defaultdict(int)
    for name in refrun.scenario_names():
        sc = Scenario(name)
        cur = Cursor(sc)
        for ev in cur.events:
            f = ev["fn"]
            if f == "run_graph_simplification":
                seen["bubbles"] += len(ev["removed"])
            elif f == "find_synteny_blocks":
                seen["unoriented"] += bool(ev["removed"])
                seen["contig_change"] += (not ev["blocks"]) or len(ev["blocks"][0]["mx"]) < len(ev["path"])
            elif f == "check_for_indels":
                seen["indel_cuts"] += len(ev["removed_edges"])
            elif f == "filter_synteny_blocks":
                seen["small_blocks"] += ev["n_in"] - len(ev["kept"])
            elif f == "mask_assemblies_with_synteny_extents":
                seen["emptied_by_slop"] += sum(len(pf["handed_to_slop"]) - len(pf["masked"]) for pf in ev["per_fasta"])
            elif f == "filter_minimizers_synteny_blocks":
                seen["filtered"] += 1
            elif f == "refine_graph":
                seen["eroded"] += len(ev["eroded_edges"])
            elif f == "merge_collinear_blocks":
                seen["merged"] += ev["n_in"] - ev["n_out"]
                for r in ev["reasons"]:
                    seen["reason_" + str(r)] += 1
        seen["warnings"] += len(sc.meta["warnings"])
    for key in ("bubbles", "unoriented", "contig_change", "indel_cuts", "small_blocks", "emptied_by_slop", "eroded", "merged", "warnings",
                "reason_id_change", "reason_ori_change", "reason_inconsistent_order", "reason_indel", "reason_merge"):
        assert seen[key] > 0, key


# ------------------------------------------------------------------------------------------------ per-function vectors
@pytest.fixture(scope="module")
def unit(golden_dir):
    with open(os.path.join(golden_dir, "unit_cases.json")) as fh:
        return json.load(fh)


def test_assembly_block_accessors(unit):
    "A:17-39 on random blocks: start / end / length / terminal / internal minimizers"
    for c in unit["assembly_block"]:
        ab = SO.AsmBlock(c["k"])
        ab.contig_id = c["contig"]
        ab.minimizers = [(h, p) for h, p in c["mx"]]
        assert (ab.start(), ab.end(), ab.length()) == (c["start"], c["end"], c["length"])
        assert [ab.contig_id, list(ab.minimizers[0]), list(ab.minimizers[-1])] == c["terminal"]
        assert [h for h, _ in ab.minimizers[1:-1]] == c["internal"]
        assert [ab.contig_id, ab.start(), ab.end()] == c["contig_start_end"]


def test_synteny_block_walk(unit):
    "B:31-46, 87-100: continue_block / start_block / extend_block along a path == the oracle's run rule (S:71-77 keeps the last run)"
    for c in unit["synteny_block_walk"]:
        eng = SO.SyntenyOracle(c["assemblies"], {}, 24, 100, [10], 10 ** 9, 1000, 0, "x")
        eng.list_mx_info = {a: {h: (v[0], v[1]) for h, v in d.items()} for a, d in c["info"].items()}
        eng.graph = SO.MxGraph()
        path = [s["mx"] for s in c["steps"]]
        # the walk's block after the last minimizer, before orientation (B:48-65 decides whether it is kept)
        cur = SO.SynBlock(24, 90, list(eng.list_mx_info))
        n = 0
        for s in c["steps"]:
            h = s["mx"]
            cont = all(info[h][0] == cur.asm[a].contig_id for a, info in eng.list_mx_info.items())
            assert cont == s["continued"]
            if not cont:
                cur = SO.SynBlock(24, 90, list(eng.list_mx_info))
                for a, info in eng.list_mx_info.items():
                    cur.asm[a].contig_id = info[h][0]
            for a, info in eng.list_mx_info.items():
                cur.asm[a].minimizers.append((h, info[h][1]))
            n = cur.n_mx()
            assert n == s["n"]
            node = cur.node(n - 1)
            assert [node[0], list(node[1])] == s["last_node"]
        got = eng._blocks_of_path(path)
        cur.orient()
        if cur.oriented():
            assert len(got) == 1
            fin = c["final"]
            for a, d in fin["asm"].items():
                assert got[0].asm[a].contig_id == d["contig"] and [list(x) for x in got[0].asm[a].minimizers] == d["mx"]
        else:
            assert got == []


def test_update_intervals(unit):
    "S:194-203: the interior (lo + 1, hi) of a block's extent, nothing when the two ends are less than 2 apart"
    for c in unit["update_intervals"]:
        eng = SO.SyntenyOracle(["a.fa.k1.w1.tsv"], {}, 24, 100, [10], 1, 1, 0, "x")
        blk = SO.SynBlock(24, 90, ["a.fa.k1.w1.tsv"])
        ab = blk.asm["a.fa.k1.w1.tsv"]
        ab.contig_id, ab.minimizers = "c", [("1", c["p1"]), ("2", c["p2"])]
        _, _, spans = eng.block_marks([blk])
        got = [list(x) + [1] for x in spans.get("a.fa.k1.w1.tsv", {}).get("c", [])]
        assert ([[1, 2, 1]] if c["pre"] else []) + got == c["out"]


def test_filter_minimizers_synteny_blocks(unit):
    "S:256-280 on random lists, black lists and interiors (half-open interval queries: the stand-in's, u11)"
    for c in unit["filter_minimizers_synteny_blocks"]:
        info = {"asm": {h: (v[0], v[1]) for h, v in c["info"].items()}}
        spans = {"asm": {"c1": [tuple(x) for x in c["intervals_c1"]]}} if c["intervals_c1"] else {"asm": {}}
        got = SO.SyntenyOracle.filter_lists({"asm": c["lists"]}, set(c["black"]), info, spans)
        assert got["asm"] == c["out"]


def test_update_list_mx_info(unit):
    for c in unit["update_list_mx_info"]:
        eng = SO.SyntenyOracle(["a", "b"], {}, 24, 100, [10], 1, 1, 0, "x")
        eng.list_mx_info = {a: {h: tuple(v) for h, v in d.items()} for a, d in c["old"].items()}
        eng.update_info(c["lists"], {a: {h: tuple(v) for h, v in d.items()} for a, d in c["new"].items()})
        assert {a: {h: list(v) for h, v in d.items()} for a, d in eng.list_mx_info.items()} == c["after"]
