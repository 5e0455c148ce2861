"""CPU restatement of ntSynt's minimizer-graph -> synteny-block stage.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product (ntsynt_amd/) never imports this module.

Follows, statement by statement where it matters for the output bytes:
  bin/ntsynt_synteny.py   (NtSyntSynteny; cited below as S:<line>)
  bin/synteny_block.py    (SyntenyBlock;  cited as B:<line>)
  bin/assembly_block.py   (AssemblyBlock; cited as A:<line>)
and, for the un-vendored ntJoin submodule (.gitmodules:1-3, directory empty in the reference tree,
SHA unknown => "parity unpinned" for these pieces): read_minimizers,
filter_minimizers, build_graph, filter_graph_global, find_paths, restated from the published
ntJoin algorithm as summarised in SURVEY.md rows C1, C2, C4, C5.

Minimizer hashes are kept as decimal strings, like the reference does, because one decision
(S:351) compares vertex names as strings.
"""
import re
import sys
from collections import defaultdict

import numpy as np

from . import nts_oracle as O

MX_SUFFIX = re.compile(r'^(\S+)\.k\d+\.w\d+.tsv')   # B:14 / S:25


# --------------------------------------------------------------------------------------------
# ntJoin pieces (rows C1, C2, C4, C5)
# --------------------------------------------------------------------------------------------
def mx_tables_from_tokens(records):
    """Row C1 (ntjoin_utils.read_minimizers): `records` = [(contig, [(hash_str, pos), ...]), ...]
    in file order.  Returns (mx_info, lists): mx_info[hash] = (contig, pos) of the FIRST sighting;
    every hash seen more than once in this assembly is dropped from both."""
    mx_info, lists, dups = {}, [], set()
    for contig, toks in records:
        if not toks:          # `if len(line) > 1` : a record without minimizers contributes no list
            continue
        lists.append([h for h, _ in toks])
        for h, pos in toks:
            if h in mx_info:
                dups.add(h)
            else:
                mx_info[h] = (contig, int(pos))
    mx_info = {h: v for h, v in mx_info.items() if h not in dups}
    lists = [[h for h in lst if h not in dups] for lst in lists]
    return mx_info, lists


def read_minimizers_tsv(path, repeat_bf=None):
    """Parse an indexlr `--long --pos [--seq]` TSV (row B4) and apply row C1.  repeat_bf (stage 3's `--filter Filter`, S:183-184,601-604;
    ntJoin's read_minimizers(file, repeat_bf), [RECALLED] like the rest of that function): a token whose k-mer text (third field) the
    filter holds is not read at all -- it is neither listed nor counted as a sighting of its hash."""
    records = []
    with open(path, encoding="utf-8") as fh:
        for line in fh:
            cols = line.strip().split("\t")
            if len(cols) > 1:
                toks = []
                for tok in cols[1].split(" "):
                    parts = tok.split(":")
                    if repeat_bf is not None and O.bf_contains(repeat_bf, O.hash_kmer(parts[2])[0]):
                        continue
                    toks.append((parts[0], int(parts[1])))
                records.append((cols[0], toks))
    return mx_tables_from_tokens(records)


def mx_records_from_arrays(names, mins):
    "Same `records` structure straight from oracle minimizer arrays (per record (h1[], pos[]))."
    return [(names[r], [(str(h), int(p)) for h, p in zip(mins[r][0].tolist(), mins[r][1].tolist())])
            for r in range(len(names))]


def filter_minimizers(list_mxs):
    "Row C2a (ntjoin_utils.filter_minimizers): keep hashes present in every assembly."
    sets = [{h for lst in list_mxs[a] for h in lst} for a in list_mxs]
    common = set.intersection(*sets) if sets else set()
    return {a: [[h for h in lst if h in common] for lst in list_mxs[a]] for a in list_mxs}


class MxGraph:
    """Undirected graph with igraph-like bookkeeping: vertices addressed by name, edges kept in
    insertion order (edge order drives S:573 and S:297), deletions preserve relative order."""

    def __init__(self):
        self.adj = {}      # name -> {neighbour name -> edge}
        self.edges = []    # edge = [s, t, weight, support(list of assemblies)] in insertion order

    def copy(self):
        g = MxGraph()
        g.edges = [[s, t, wgt, list(sup)] for s, t, wgt, sup in self.edges]
        g.adj = {v: {} for v in self.adj}
        for e in g.edges:
            g.adj[e[0]][e[1]] = e
            g.adj[e[1]][e[0]] = e
        return g

    def add_vertex(self, v):
        self.adj.setdefault(v, {})

    def add_edge(self, s, t, weight, support):
        old = self.adj[s].get(t)
        if old is not None:           # see module docstring of tests: never observed; overwrite
            old[2], old[3] = weight, support
            return old
        e = [s, t, weight, support]
        self.edges.append(e)
        self.adj[s][t] = e
        self.adj[t][s] = e
        return e

    def degree(self, v):
        return len(self.adj[v])

    def delete_edges(self, dead):
        dead_ids = {id(e) for e in dead}
        if not dead_ids:
            return
        for e in dead:
            self.adj[e[0]].pop(e[1], None)
            self.adj[e[1]].pop(e[0], None)
        self.edges = [e for e in self.edges if id(e) not in dead_ids]

    def delete_vertices(self, names):
        names = set(names)
        if not names:
            return
        for v in names:
            if v in self.adj:
                for u in list(self.adj[v]):
                    self.adj[u].pop(v, None)
                del self.adj[v]
        self.edges = [e for e in self.edges if e[0] not in names and e[1] not in names]


def build_graph(list_mxs, weights, graph=None, black_list=None):
    """Row C2b (ntjoin_utils.build_graph).  Edges are created in the order a dict-of-dicts
    `edges[source][target]` would list them: grouped by source in order of first use as a source,
    then by insertion (SURVEY.md hard part H4).  With `graph` given, vertices/edges are added in
    place (u8); names in `black_list` already exist and are not re-added."""
    if graph is None:
        graph = MxGraph()
    black_list = black_list or set()
    vertices = []
    seen_v = set()
    edges = defaultdict(dict)
    for asm in list_mxs:
        for lst in list_mxs[asm]:
            for a, b in zip(lst, lst[1:]):
                if a in edges and b in edges[a]:
                    edges[a][b].append(asm)
                elif b in edges and a in edges[b]:
                    edges[b][a].append(asm)
                else:
                    edges[a][b] = [asm]
                if a not in seen_v:
                    seen_v.add(a)
                    vertices.append(a)
            if lst and lst[-1] not in seen_v:
                seen_v.add(lst[-1])
                vertices.append(lst[-1])
    for v in vertices:
        if v not in black_list:
            graph.add_vertex(v)
    for s in edges:
        for t in edges[s]:
            graph.add_vertex(s)
            graph.add_vertex(t)
            sup = edges[s][t]
            graph.add_edge(s, t, sum(weights[a] for a in sup), sup)
    return graph


def filter_graph_global(graph, n, weights):
    "Row C4 (Ntjoin.filter_graph_global): drop edges lighter than n; returns a new graph (u9)."
    if n <= min(weights.values()):
        return graph
    g = graph.copy()
    g.delete_edges([e for e in g.edges if e[2] < n])
    return g


def find_paths(graph, ref_mx_info):
    """Row C5 (Ntjoin.find_paths / ntjoin_find_paths): one ordered list of hash names per connected
    component that is a simple path (exactly two degree-1 vertices, the rest degree 2).  The walk
    starts at the end whose position in the reference assembly (last of the descending-sorted
    file list, i.e. lexicographically smallest) is smaller."""
    paths, seen = [], set()
    for v0 in graph.adj:
        if v0 in seen:
            continue
        comp, stack = [], [v0]
        seen.add(v0)
        while stack:
            v = stack.pop()
            comp.append(v)
            for u in graph.adj[v]:
                if u not in seen:
                    seen.add(u)
                    stack.append(u)
        if len(comp) < 2:
            continue
        ends = [v for v in comp if graph.degree(v) == 1]
        if len(ends) != 2 or any(graph.degree(v) > 2 for v in comp):
            continue
        a, b = ends
        src = a if ref_mx_info[a][1] <= ref_mx_info[b][1] else b
        if ref_mx_info[a][1] == ref_mx_info[b][1]:
            src = b           # ties: `[... == min_pos].pop()` takes the last listed end
        path, prev, cur = [src], None, src
        while True:
            nxt = [u for u in graph.adj[cur] if u != prev]
            if not nxt:
                break
            prev, cur = cur, nxt[0]
            path.append(cur)
        if len(path) == len(comp):
            paths.append(path)
    return paths


# --------------------------------------------------------------------------------------------
# Block model (rows C6, C7, C10)
# --------------------------------------------------------------------------------------------
class AsmBlock:
    "A:1-43"

    def __init__(self, k):
        self.contig_id = None
        self.minimizers = []     # [(hash_str, pos)]
        self.ori = None
        self.k = k

    def start(self):
        return min(self.minimizers[0][1], self.minimizers[-1][1])          # A:17-19

    def end(self):
        return max(self.minimizers[0][1], self.minimizers[-1][1]) + self.k  # A:21-23

    def length(self):
        return self.end() - self.start()

    def shallow(self):
        c = AsmBlock(self.k)
        c.contig_id, c.minimizers, c.ori = self.contig_id, self.minimizers, self.ori
        return c


class SynBlock:
    "B:16-116"

    def __init__(self, k, m, assemblies):
        self.asm = {a: AsmBlock(k) for a in assemblies}
        self.m = m
        self.broken_reason = None

    def n_mx(self):
        return len(self.asm[list(self.asm)[-1]].minimizers)      # B:97-100

    def orient(self):                                             # B:48-65
        for blk in self.asm.values():
            pos = [p for _, p in blk.minimizers]
            steps = list(zip(pos, pos[1:]))
            if all(x < y for x, y in steps):
                blk.ori = "+"
            elif all(x > y for x, y in steps):
                blk.ori = "-"
            else:
                up = [x < y for x, y in steps]
                pos_perc = up.count(True) / float(len(pos) - 1) * 100
                neg_perc = 100 - pos_perc
                blk.ori = "+" if pos_perc >= self.m else ("-" if neg_perc >= self.m else "?")

    def oriented(self):                                           # B:68-70
        return all(blk.ori in ("+", "-") for blk in self.asm.values())

    def node(self, i):                                            # B:87-95
        return self.asm[sorted(self.asm)[0]].minimizers[i][0], \
            [self.asm[a].minimizers[i][1] for a in sorted(self.asm)]

    def text(self, num, verbose=False):                           # B:72-85
        out = []
        for a in sorted(self.asm):
            blk = self.asm[a]
            m = MX_SUFFIX.search(a)
            name = m.group(1) if m else a
            row = f"{num}\t{name}\t{blk.contig_id}\t{blk.start()}\t{blk.end()}\t{blk.ori}\t{len(blk.minimizers)}"
            if verbose:
                row = f"{row.strip()}\t{self.broken_reason}"
            out.append(row + "\n")
        return "".join(out)

    def sort_key(self):                                           # B:102-109
        blk = self.asm[sorted(self.asm)[0]]
        return (blk.contig_id, blk.start())

    def long_enough(self, z):
        return all(blk.length() >= z for blk in self.asm.values())


# --------------------------------------------------------------------------------------------
# The synteny engine (rows C3, C6, C8, C9, C11, C12 and the driver S:593-647)
# --------------------------------------------------------------------------------------------
class SyntenyOracle:
    """files: minimizer TSV names (identify assemblies; S:34 sorts them descending);
    genomes: {tsv name: oracle Genome} for the refinement re-sketch (S:134-192);
    bf: common Bloom filter (uint8 array) or None."""

    def __init__(self, files, genomes, k, w, w_rounds, bp, collinear_merge, z, prefix, bf=None,
                 simplify=True, m=90, n=0, threads=1, log=None, interarrivals=False):
        self.interarrivals = interarrivals                        # --interarrivals (S:557-564, S:626-627)
        self.files = sorted(files, reverse=True)                  # S:34
        self.genomes = genomes
        self.k, self.w, self.w_rounds = k, w, list(w_rounds)
        self.bp, self.z, self.prefix, self.m = bp, z, prefix, m
        self.bf, self.simplify, self.threads = bf, simplify, threads
        self.n = n or len(self.files)                             # S:46-47
        cm = str(collinear_merge)
        if mt := re.search(r"^(\d+)w$", cm):                      # S:37-42
            self.collinear_merge = int(mt.group(1)) * w
        elif mt := re.search(r"^(\d+)$", cm):
            self.collinear_merge = int(mt.group(1))
        else:
            raise ValueError("--collinear-merge must be provided with an integer value or string in the form '<num>w'")
        self.weights = {f: 1 for f in self.files}                 # S:32
        self.list_mx_info = {}
        self.list_mxs = {}
        self.graph = None
        self.log = log or (lambda *a: None)
        self.outputs = {}      # file name -> text (also written to disk by main())

    # -- graph simplification: S:548-590 (row C3) ---------------------------------------------
    def _partially_anchored(self, graph, v, wmax):
        return [e[2] for e in graph.adj[v].values()].count(wmax) == 1

    def simplify_graph(self, graph):
        wmax = sum(self.weights.values())
        doomed = []
        for e in graph.edges:
            s, t = e[0], e[1]
            if graph.degree(s) == 3 and graph.degree(t) == 3 and \
                    self._partially_anchored(graph, s, wmax) and self._partially_anchored(graph, t, wmax):
                common = [u for u in graph.adj[s] if u != t and u in graph.adj[t]]
                if len(common) == 1:            # direct edge + exactly one 2-step path  (S:581-582)
                    doomed.append(common[0])
                    e[2] = wmax                 # in-loop mutation, visible to later edges (S:586)
        g = graph.copy()
        g.delete_vertices(doomed)
        return g

    # -- path -> blocks: S:66-106 (row C6) -------------------------------------------------------
    def _blocks_of_path(self, path):
        out, drop = [], []
        cur = SynBlock(self.k, self.m, list(self.list_mx_info))
        for h in path:
            if all(info[h][0] == cur.asm[a].contig_id for a, info in self.list_mx_info.items()):
                for a, info in self.list_mx_info.items():
                    cur.asm[a].minimizers.append((h, info[h][1]))
            else:
                # `past_start_flag` is never set (S:71,77): an earlier partial block is dropped
                cur = SynBlock(self.k, self.m, list(self.list_mx_info))
                for a, info in self.list_mx_info.items():
                    cur.asm[a].contig_id = info[h][0]
                    cur.asm[a].minimizers.append((h, int(info[h][1])))
        cur.orient()
        if cur.oriented():
            out.append(cur)
        else:
            drop.extend(h for h, _ in cur.asm[list(cur.asm)[-1]].minimizers)
        if drop:
            g = self.graph.copy()
            g.delete_vertices(drop)
            self.graph = g
        return out

    def blocks_of_paths(self, paths):                             # S:543-546
        return [b for p in paths for b in self._blocks_of_path(p)]

    # -- indel split: S:364-409 (row C8) ------------------------------------------------------------
    def split_indels(self, blocks):
        out, dead = [], []
        for blk in blocks:
            cuts = []
            for i in range(blk.n_mx() - 1):
                h1, p1 = blk.node(i)
                h2, p2 = blk.node(i + 1)
                gaps = [abs(x - y) for x, y in zip(p1, p2)]
                if max(gaps) - min(gaps) > self.bp:
                    cuts.append(i + 1)
                    e = self.graph.adj[h1].get(h2)
                    if e is not None:
                        dead.append(e)
            if not cuts:
                out.append(blk)
                continue
            bounds = [0] + cuts + [blk.n_mx()]                    # S:370-388
            for lo, hi in zip(bounds, bounds[1:]):
                nb = SynBlock(self.k, self.m, list(blk.asm))
                for a in blk.asm:
                    piece = blk.asm[a].shallow()
                    piece.minimizers = blk.asm[a].minimizers[lo:hi]
                    nb.asm[a] = piece
                out.append(nb)
        g = self.graph.copy()                                     # remove_flagged_edges
        g.delete_edges([g.adj[e[0]][e[1]] for e in dead if e[1] in g.adj.get(e[0], {})])
        self.graph = g
        return out

    # -- small-block filter: S:411-426 (row C9) ------------------------------------------------------
    def drop_small(self, blocks, min_mx):
        keep, drop = [], []
        for blk in blocks:
            if all(len(ab.minimizers) >= min_mx for ab in blk.asm.values()):
                keep.append(blk)
            else:
                drop.extend(h for h, _ in blk.asm[list(blk.asm)[-1]].minimizers)
        g = self.graph.copy()
        g.delete_vertices(drop)
        self.graph = g
        return keep

    # -- collinear merge: S:428-472 (row C12) --------------------------------------------------------
    @staticmethod
    def _gap(b1, b2):
        if b1.ori == "-" and b2.ori == "-":
            return b1.start() - b2.end()
        return b2.start() - b1.end()

    def merge_collinear(self, blocks):
        out = []
        cur = blocks[0]
        for blk in blocks[1:]:
            same_ori = same_ctg = True
            diffs = []
            for a, ab in cur.asm.items():
                if ab.ori != blk.asm[a].ori:
                    same_ori = False
                if ab.contig_id != blk.asm[a].contig_id:
                    same_ctg = False
                diffs.append(self._gap(ab, blk.asm[a]))
            spread = max(diffs) - min(diffs)
            if (not same_ori) or (not same_ctg) or spread > self.bp - self.k or \
                    max(diffs) >= self.collinear_merge:
                if not same_ctg:
                    blk.broken_reason = "id_change"
                elif not same_ori:
                    blk.broken_reason = "ori_change"
                elif any(d < 0 for d in diffs):
                    blk.broken_reason = "inconsistent_order"
                elif spread > self.bp - self.k:
                    blk.broken_reason = "indel"
                elif max(diffs) >= self.collinear_merge:
                    blk.broken_reason = "merge"
                out.append(cur)
                cur = blk
            else:
                for a, ab in blk.asm.items():
                    cur.asm[a].minimizers.extend(ab.minimizers)
        out.append(cur)
        return out

    # -- refinement helpers: S:118-290 (rows B5, C11) ---------------------------------------------------
    def mask_intervals(self, blocks, w):
        """S:118-157 without files: per assembly, per contig, the [start,end) intervals to hard-mask:
        block extents longer than max(2w, w+k+1), shrunk by (w+k) on both sides (bedtools slop with
        negative values), clipped to the contig; empty results mask nothing (u10)."""
        masks = {}
        for blk in blocks:
            for a, ab in blk.asm.items():
                s, e = ab.start(), ab.end()
                if e - s > max(2 * w, w + self.k + 1):
                    s2, e2 = s + (w + self.k), e - (w + self.k)
                    if e2 > s2:
                        masks.setdefault(a, {}).setdefault(ab.contig_id, []).append((s2, e2))
        return masks

    def masked_genome(self, asm, ctg_masks):
        g = self.genomes[asm]
        seqs = []
        for i, name in enumerate(g.names):
            rec = g.record(i)
            if name in ctg_masks:
                buf = bytearray(rec)
                for s, e in ctg_masks[name]:
                    s, e = max(0, s), min(len(buf), e)
                    if e > s:
                        buf[s:e] = b"N" * (e - s)
                rec = bytes(buf)
            seqs.append(rec)
        return O.Genome(g.names, seqs)

    def sketch_masked(self, asm, ctg_masks, new_w):                 # S:134-192: maskfasta + indexlr + read_minimizers
        "(mx_info, lists) of assembly `asm` re-sketched with hard masks (a seam: list-level tests script this step)"
        mg = self.masked_genome(asm, ctg_masks)
        mins = O.minimize(mg, self.k, new_w, self.bf, self.threads, repeat=getattr(self, "refine_repeat", None))   # S:172-180: --filter Indexlr adds -r
        records = mx_records_from_arrays(mg.names, mins)
        screen = getattr(self, "screen_repeat", None)                # S:183-184: --filter Filter, read_minimizers(file, repeat_bf)
        if screen is not None:
            g = self.genomes[asm]                                     # (a minimizer's k-mer holds no masked base: the unmasked record serves)
            kept = []
            for r, (name, toks) in enumerate(records):
                rec = g.record(r)
                kept.append((name, [(h, p) for h, p in toks if not O.bf_contains(screen, O.hash_kmer(rec[p:p + self.k])[0])]))
            records = kept
        return mx_tables_from_tokens(records)

    def block_marks(self, blocks):                                 # S:205-226 (find_mx_in_blocks) + S:194-203 (update_intervals)
        "(terminal minimizers, internal minimizers, {assembly: {contig: [(lo + 1, hi)]}}) of the blocks"
        terminal, internal, spans = set(), set(), defaultdict(dict)
        for blk in blocks:
            for a, ab in blk.asm.items():
                first, last = ab.minimizers[0], ab.minimizers[-1]
                terminal.add(first[0])
                terminal.add(last[0])
                lo, hi = min(first[1], last[1]), max(first[1], last[1])
                if hi - lo >= 2:                                     # S:199
                    spans[a].setdefault(ab.contig_id, []).append((lo + 1, hi))
                internal.update(h for h, _ in ab.minimizers[1:-1])
        return terminal, internal, spans

    @staticmethod
    def filter_lists(list_mxs, internal, new_info, spans):          # S:256-280 (filter_minimizers_synteny_blocks)
        """Minimizers of the re-sketch that are neither internal to a block nor inside a block's interior; a list is cut where the
        stretch between two kept neighbours reaches into an interior.  Interval queries are half-open (u11)."""
        idx = {a: {c: _IntervalSet(v) for c, v in d.items()} for a, d in spans.items()}
        filt = {}
        for a in list_mxs:
            out_lists = []
            for lst in list_mxs[a]:
                cur = []
                for h in lst:
                    ctg, pos = new_info[a][h]
                    iv = idx.get(a, {}).get(ctg)
                    if cur and iv is not None:
                        prev_pos = new_info[a][cur[-1]][1]
                        if iv.overlaps(min(prev_pos, pos), max(prev_pos, pos)):
                            out_lists.append(cur)
                            cur = []
                    if h not in internal and (iv is None or not iv.overlaps(pos, pos + 1)):
                        cur.append(h)
                out_lists.append(cur)
            filt[a] = out_lists
        return filt

    def update_info(self, filt, new_info):                          # S:282-290 (update_list_mx_info)
        valid = {h for ls in filt.values() for lst in ls for h in lst}
        for a, info in new_info.items():
            for h in info:
                if h in valid:
                    self.list_mx_info[a][h] = info[h]

    def new_minimizers(self, blocks, new_w, prev_w):               # S:532-541
        masks = self.mask_intervals(blocks, prev_w)
        list_mxs, new_info = {}, {}
        # S:138 iterates the assemblies that appear in synteny_beds (block.assembly_blocks order)
        order = list(blocks[0].asm) if blocks else []
        for a in order:
            new_info[a], list_mxs[a] = self.sketch_masked(a, masks.get(a, {}), new_w)
        terminal, internal, spans = self.block_marks(blocks)
        filt = self.filter_lists(list_mxs, internal, new_info, spans)
        filt = filter_minimizers(filt)                                  # S:539
        self.update_info(filt, new_info)
        return filt, terminal

    # -- last-round erosion: S:292-362 (row C12) ------------------------------------------------------
    def _too_close(self, a, b):                                         # S:305-310
        return any(abs(info[a][1] - info[b][1]) < self.k for info in self.list_mx_info.values())

    def _erode(self, source, target):                                   # S:312-340
        erode_target = True
        cs, ct = source, target
        dead, visited = {}, {cs, ct}
        sname, tname = source, target
        while self._too_close(sname, tname):
            v = ct if erode_target else cs
            for e in self.graph.adj[v].values():
                dead[id(e)] = e
            nb = [u for u in self.graph.adj[v] if u not in visited]
            if not nb:
                break
            assert len(nb) == 1
            if erode_target:
                ct = tname = nb[0]
                erode_target = False
                visited.add(ct)
            else:
                cs = sname = nb[0]
                erode_target = True
                visited.add(cs)
        return list(dead.values())

    def refine_graph(self, flagged):                                    # S:343-362
        if not flagged:
            return self.graph
        dead = []
        for s, t in flagged:
            if s > t:                      # string comparison of decimal names (S:351)
                s, t = t, s
            if self.graph.degree(s) != 1 or self.graph.degree(t) != 1:
                continue
            dead.extend(self._erode(s, t))
        if not dead:
            return self.graph
        g = self.graph.copy()
        g.delete_edges([g.adj[e[0]][e[1]] for e in dead if e[1] in g.adj.get(e[0], {})])
        return g

    # -- writers: S:496-503, 516-523, 634-641 (row C10) -------------------------------------------------
    def check_non_overlapping(self, blocks):                             # S:234-253 (--dev)
        """Final self-check of developer mode: a warning on stderr for every block whose extent overlaps an earlier block's
        extent in the same assembly and contig by at least z.  (An intervaltree query [start:end) against the extents
        inserted so far; blocks below the length filter neither warn nor count.)"""
        seen = defaultdict(list)                                         # (assembly, contig) -> [(start, end)]
        for blk in blocks:
            for a, ab in blk.asm.items():
                if not all(x.length() >= self.z for x in blk.asm.values()):
                    continue
                start, end = ab.start(), ab.end()
                for s0, e0 in seen[(a, ab.contig_id)]:
                    if s0 < end and start < e0 and min(end, e0) - max(start, s0) >= self.z:
                        print("WARNING: detected overlapping segments for this block:", a, ab.contig_id, start, end, "\n",
                              file=sys.stderr, flush=True)
                        break
                seen[(a, ab.contig_id)].append((start, end))

    def _emit(self, name, blocks, verbose=False):
        rows, num = [], 0
        for blk in blocks:
            if not blk.long_enough(self.z):
                continue
            rows.append(blk.text(num, verbose))
            num += 1
        text = "".join(rows)
        self.outputs[name] = text
        with open(name, "w", encoding="utf-8") as fh:
            fh.write(text)

    # -- S:476-530 ------------------------------------------------------------------------------------------
    def refine(self, blocks):
        prev_w = self.w
        for new_w in self.w_rounds:
            self.log(f"refining with w={new_w}")
            new_lists, terminal = self.new_minimizers(blocks, new_w, prev_w)
            graph = build_graph(new_lists, self.weights, graph=self.graph, black_list=terminal)
            if self.simplify:
                self.graph = self.simplify_graph(self.graph)   # promotions land in `graph` too (S:485)
            last = new_w == self.w_rounds[-1]
            if last:
                flagged = [(e[0], e[1]) for e in graph.edges if e[2] < self.n]      # S:292-303
                g = graph.copy()
                g.delete_edges([e for e in g.edges if e[2] < self.n])
                self.graph = g
                self.graph = self.refine_graph(flagged)
            else:
                self.graph = filter_graph_global(graph, self.n, self.weights)
            blocks = self.blocks_of_paths(find_paths(self.graph, self.list_mx_info[self.files[-1]]))
            blocks = self.split_indels(blocks)
            blocks = self.drop_small(blocks, 4)
            ordered = sorted(blocks, key=SynBlock.sort_key)
            self._emit(f"{self.prefix}.pre-collinear-merge.synteny_blocks.tsv", ordered)
            if last:
                # S:505-510: both calls are unconditional -- a last round that leaves no block, or none of at least z bases, ends the
                # reference with `IndexError: list index out of range` at S:437 (`curr_block = blocks[0]`); so does merge_collinear
                merged = self.merge_collinear(ordered)
                merged = [b for b in merged if b.long_enough(self.z)]
                merged = self.merge_collinear(merged)
                if getattr(self, "dev", False):                           # S:513-514
                    self.check_non_overlapping(merged)
                self._emit(f"{self.prefix}.synteny_blocks.tsv", merged, verbose=True)
            prev_w = new_w
        return blocks

    # -- S:593-647 ------------------------------------------------------------------------------------------
    def load(self, tables):
        "tables: {tsv name: (mx_info, lists)} as produced by row C1; kept in self.files order."
        for f in self.files:
            self.list_mx_info[f], self.list_mxs[f] = tables[f]

    def main(self):
        if len(self.w_rounds) != len(set(self.w_rounds)):                  # S:597-599
            print("Error: duplicate values found in w_rounds!", file=sys.stderr)
            sys.exit(1)
        self.list_mxs = filter_minimizers(self.list_mxs)                    # S:612
        self.graph = build_graph(self.list_mxs, self.weights)
        if self.simplify:
            self.graph = self.simplify_graph(self.graph)                    # S:615-616
        self.graph = filter_graph_global(self.graph, self.n, self.weights)  # S:617
        paths = find_paths(self.graph, self.list_mx_info[self.files[-1]])   # S:620
        blocks = self.blocks_of_paths(paths)
        blocks = self.split_indels(blocks)
        blocks = self.drop_small(blocks, 4)
        if self.interarrivals:                                              # S:557-564: one distance per line, blocks in path order
            lines = []
            for blk in blocks:
                for ab in blk.asm.values():
                    lines.extend(str(abs(b[1] - a[1])) for a, b in zip(ab.minimizers, ab.minimizers[1:]))
            self.outputs[f"{self.prefix}.interarrivals.tsv"] = "".join(x + "\n" for x in lines)
            with open(f"{self.prefix}.interarrivals.tsv", "w", encoding="utf-8") as fout:
                fout.write(self.outputs[f"{self.prefix}.interarrivals.tsv"])
        ordered = sorted(blocks, key=SynBlock.sort_key)
        self.initial_blocks = ordered
        if not ordered:
            print("Error - no paths found. Try adjusting the specified k/w parameters.")
            sys.exit(1)
        self._emit(f"{self.prefix}.synteny_blocks.tsv", ordered)
        self.refine(blocks)
        return self.outputs


class _IntervalSet:
    """Half-open interval overlap queries (stands in for ncls.NCLS.has_overlap, u11)."""

    def __init__(self, ivs):
        ivs = sorted(ivs)
        self.starts = np.array([s for s, _ in ivs], dtype=np.int64)
        ends = np.array([e for _, e in ivs], dtype=np.int64)
        self.maxend = np.maximum.accumulate(ends) if len(ends) else ends

    def overlaps(self, s, e):
        "any stored [a,b) with a < e and b > s"
        if e <= s:
            return False
        i = int(np.searchsorted(self.starts, e, side="left"))   # intervals with start < e
        return i > 0 and int(self.maxend[i - 1]) > s


# --------------------------------------------------------------------------------------------
# End-to-end driver: what `ntSynt` + the Snakemake rules do, in process
# --------------------------------------------------------------------------------------------
def divergence_defaults(d):
    "bin/ntSynt:89-99 -> (indel, merge, w_rounds, block_size)"
    if d < 1:
        return 10000, 10000, [100, 10], 500
    if d <= 10:
        return 50000, 100000, [250, 100], 1000
    if d <= 100:
        return 100000, 1000000, [500, 250], 10000
    raise ValueError("--divergence must be a value between 0 and 100")


def run_pipeline(fastas, k=24, w=1000, fpr=0.025, prefix=None, w_rounds=(100, 10), indel=10000,
                 merge=10000, block_size=500, common=True, simplify=True, threads=1,
                 write_mx_tsv=True, log=None, bf_rounding="up", interarrivals=False, repeat=False, n=0, m=90):
    """FASTA paths -> {output file name: text}; files are written into the CWD like the reference.
    Stage order: make_common_bf (smk:55-62) -> indexlr per genome (smk:74-85) -> ntsynt_run.py
    (smk:87-103)."""
    import os
    prefix = prefix or f"ntSynt.k{k}.w{w}"
    genomes = {p: O.read_fasta(p) for p in fastas}
    bf = O.common_bf(genomes, k, fpr, threads, rounding=bf_rounding) if common else None
    rep = None
    if repeat:      # smk:65-85 (experimental): repeat filter over all genomes, sized from the first; used by the whole-genome indexlr only
        import math
        size_bits = math.ceil((-1 * genomes[fastas[0]].total_bp) / (math.log(1 - fpr)))     # ntsynt_make_repeat_bfs.py:25-34
        rep = O.repeat_bf([genomes[p] for p in fastas], k, (int(size_bits / 8) + 7) // 8 * 8)
    tables, by_tsv = {}, {}
    for p in fastas:
        tsv = f"{os.path.basename(p)}.k{k}.w{w}.tsv"
        mins = O.minimize(genomes[p], k, w, bf, threads, repeat=rep)
        if write_mx_tsv:
            O.write_indexlr_tsv(tsv, genomes[p], mins, k)
            tables[tsv] = read_minimizers_tsv(tsv)
        else:
            tables[tsv] = mx_tables_from_tokens(mx_records_from_arrays(genomes[p].names, mins))
        by_tsv[tsv] = genomes[p]
    eng = SyntenyOracle(list(tables), by_tsv, k, w, w_rounds, indel, merge, block_size, prefix,
                        bf=bf, simplify=simplify, threads=threads, log=log, interarrivals=interarrivals, n=n, m=m)
    eng.load(tables)
    eng.main()
    eng.bf = bf
    return eng
