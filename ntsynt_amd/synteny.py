"""Minimizer graph -> synteny blocks on the host side of the HIP path.

This is the array-based counterpart of the reference's graph stage (bin/ntsynt_synteny.py,
bin/synteny_block.py, bin/assembly_block.py; ntJoin's graph helpers): vertices are integer ids,
hashes/positions live in numpy arrays, edges in flat arrays kept in the reference's edge order.
The heavy joins (duplicate removal, cross-assembly intersection, adjacency edges: rows C1, C2) are
done by `graph_fn` (the GPU: ntsynt_amd.graph.build_graph_device) and the re-sketch of masked
genomes (row B5) by `sketch_fn` (the GPU: nts_sketch); this module holds the rules that decide the
output bytes.  It never touches the oracle.

Reference behaviours reproduced on purpose (cited as S:<line> = bin/ntsynt_synteny.py):
  * edge order = dict-of-dicts order of ntJoin's build_graph (drives S:573-586 and S:297-300);
  * in refinement rounds bubble removal only promotes weights, its vertex deletions are lost (S:483-491);
  * a path whose contig changes keeps only its last run (S:71-77);
  * vertex names compare as decimal strings when normalising flagged pairs (S:351).
"""
import os
import re
import sys
from dataclasses import dataclass, field

import numpy as np

MX_SUFFIX = re.compile(r'^(\S+)\.k\d+\.w\d+.tsv')
EMPTY_FINAL = ("list index out of range (the last refinement round left no synteny block{}: the reference stops here too -- "
               "merge_collinear_blocks takes blocks[0] of an empty list, bin/ntsynt_synteny.py:437)")


@dataclass
class GraphArrays:
    """Result of one graph build (mirror of nts_graph in include/ntsynt_hip.h)."""
    v_hash: np.ndarray                      # [nv] uint64, ascending
    occ_rec: np.ndarray                     # [G, nv] int64
    occ_pos: np.ndarray                     # [G, nv] int64
    e_u: np.ndarray                         # [ne] int64, orientation of first sighting
    e_v: np.ndarray
    e_w: np.ndarray                         # [ne] int64
    e_first: np.ndarray                     # [ne] int64 sequence number of first sighting
    dict_ordered: bool = False              # edges already in the order dict_order() would give (nts_graph_build)


def dict_order(e_u, e_first, nv):
    """Permutation putting edges in the order `[(s, t) for s in edges for t in edges[s]]` yields:
    sources by the time they first became a source, then by creation time."""
    if e_u.size == 0:
        return np.zeros(0, dtype=np.int64)
    src_rank = np.full(nv, np.iinfo(np.int64).max, dtype=np.int64)
    np.minimum.at(src_rank, e_u, e_first)
    return np.lexsort((e_first, src_rank[e_u]))


@dataclass
class Block:
    vids: np.ndarray                        # vertex ids in path order
    rec: list                               # per assembly: contig (record index)
    ori: list                               # per assembly: '+', '-', '?'
    reason: object = None
    # after collinear merging only the ends and the count matter
    first_pos: list = field(default_factory=list)
    last_pos: list = field(default_factory=list)
    n_mx: int = 0


class SyntenyEngine:
    """files: minimizer-TSV names identifying the assemblies (any order; sorted descending like
    S:34); contig_names[a]: record names of assembly a (indexable by record id);
    graph_fn(lists, keep, list_ids) -> GraphArrays with lists[a] = (h1, rec, pos) arrays;
    sketch_fn(a, masks, w) -> (h1, rec, pos) of assembly a re-sketched with hard masks [(rec, s, e)]; if it carries an
    attribute all_at_once({a: masks}, w) -> {a: (h1, rec, pos)}, a refinement round calls that once instead;
    walk_fn / scan_fn: chain walk and per-path scan (native host helpers nts_walk_chains / nts_path_scan)."""

    def __init__(self, files, contig_names, k, w, w_rounds, bp, collinear_merge, z, prefix, graph_fn, sketch_fn,
                 walk_fn, simplify=True, m=90, n=0, log=None, scan_fn=None, degree_fn=None, dev=False, interarrivals=False):
        order = sorted(range(len(files)), key=lambda i: files[i], reverse=True)
        self.input_order = order                       # engine index a -> caller's assembly index
        self.files = [files[i] for i in order]
        self.contigs = [contig_names[i] for i in order]
        self.G = len(files)
        self.k, self.w, self.w_rounds = k, w, list(w_rounds)
        self.bp, self.z, self.prefix, self.m = bp, z, prefix, m
        self.simplify = simplify
        self.dev = dev                                 # --dev: overlap self-check of the final blocks (S:513-514)
        self.interarrivals = interarrivals             # --interarrivals: distances between neighbouring minimizers of the initial blocks (S:557-564)
        self.n = n or self.G
        cm = str(collinear_merge)
        if mt := re.search(r"^(\d+)w$", cm):
            self.collinear_merge = int(mt.group(1)) * w
        elif mt := re.search(r"^(\d+)$", cm):
            self.collinear_merge = int(mt.group(1))
        else:
            raise ValueError("--collinear-merge must be provided with an integer value or string in the form '<num>w'")
        self.graph_fn, self.sketch_fn, self.walk_fn = graph_fn, sketch_fn, walk_fn
        if scan_fn is None:
            from .graph import scan_paths as scan_fn       # native host helper (nts_path_scan)
        self.scan_fn = scan_fn
        self.degree_fn = degree_fn
        self._asm_names = None
        self.times = {}
        if os.environ.get("NTS_ENGINE_TIMES"):             # wall clock per step, for scripts/e2e_run.py
            self._instrument()
        self.log = log or (lambda *a: None)
        self.outputs = {}
        self.stats = {"bubbles": 0, "unoriented": 0, "indel_cuts": 0, "small_blocks": 0, "merged": 0, "eroded_edges": 0}
        self.ref = self.G - 1                          # lexicographically smallest file (S:34 + ntJoin)
        self.out_order = sorted(range(self.G), key=lambda a: self.files[a])
        # graph state
        self.v_hash = np.zeros(0, np.uint64)
        self.v_alive = np.zeros(0, bool)
        self.v_rec = np.zeros((self.G, 0), np.int64)
        self.v_pos = np.zeros((self.G, 0), np.int64)
        self.e_u = np.zeros(0, np.int64)
        self.e_v = np.zeros(0, np.int64)
        self.e_w = np.zeros(0, np.int64)
        self.e_alive = np.zeros(0, bool)
        self._hs = np.zeros(0, np.uint64)                  # live-vertex index: hashes ascending, and their ids
        self._hid = np.zeros(0, np.int64)

    def _instrument(self):
        import time

        def wrap(name, fn):
            def timed(*a, **k):
                t0 = time.perf_counter()
                try:
                    return fn(*a, **k)
                finally:
                    self.times[name] = self.times.get(name, 0.0) + time.perf_counter() - t0
            timed.oriented = getattr(fn, "oriented", False)
            if getattr(fn, "all_at_once", None) is not None:
                timed.all_at_once = wrap(name, fn.all_at_once)
            return timed
        for name in ("graph_fn", "sketch_fn", "walk_fn", "scan_fn") + (("degree_fn",) if self.degree_fn else ()):
            setattr(self, name, wrap(name, getattr(self, name)))
        for name in ("_add_graph", "_simplify", "_degrees", "_paths", "_blocks_of_paths", "_drop_small", "_new_round_graph",
                     "_refine_graph", "_sorted", "_emit", "_merge", "_find_edges", "_live_index", "_delete_vertices"):
            setattr(self, name, wrap(name, getattr(self, name)))

    # ------------------------------------------------------------------ graph bookkeeping
    def _degrees(self):
        if self.degree_fn is not None:                         # nts_edge_degrees (saturates at 255; users ask == 1 / == 3)
            return self.degree_fn(self.v_hash.size, self.e_u, self.e_v, self.e_alive)
        m = self.e_alive
        n = self.v_hash.size
        return np.bincount(self.e_u[m], minlength=n) + np.bincount(self.e_v[m], minlength=n)

    def _delete_vertices(self, vids):
        if len(vids) == 0:
            return
        dead = np.zeros(self.v_hash.size, bool)
        dead[np.asarray(vids, dtype=np.int64)] = True
        self.v_alive &= ~dead
        self.e_alive &= ~(dead[self.e_u] | dead[self.e_v])

    def _live_index(self):
        """(hashes ascending, vertex ids) of the live vertices.  The index is kept across rounds: deleted vertices
        drop out here, new ones are merged in by _add_graph; two live vertices never share a hash."""
        keep = self.v_alive[self._hid]
        if not keep.all():
            self._hs, self._hid = self._hs[keep], self._hid[keep]
        return self._hs, self._hid

    def _find_edges(self, us, vs):
        """Indices of the live edges {us[i], vs[i]} (queries without an edge are skipped).  Queries are few next to
        the edge list (indel cuts, edges between re-used vertices), so the edges touching a queried vertex are
        filtered out in one pass and only those are sorted."""
        us, vs = np.asarray(us, np.int64), np.asarray(vs, np.int64)
        if us.size == 0 or self.e_u.size == 0:
            return np.zeros(0, np.int64), np.zeros(us.size, bool)
        touched = np.zeros(self.v_hash.size, bool)
        touched[us] = True
        idx = np.flatnonzero((touched[self.e_u] | touched[self.e_v]) & self.e_alive)
        if idx.size == 0:
            return np.zeros(0, np.int64), np.zeros(us.size, bool)
        key = (np.minimum(self.e_u[idx], self.e_v[idx]) << 32) | np.maximum(self.e_u[idx], self.e_v[idx])
        srt = np.argsort(key)
        key, idx = key[srt], idx[srt]
        q = (np.minimum(us, vs) << 32) | np.maximum(us, vs)
        p = np.minimum(np.searchsorted(key, q), key.size - 1)
        ok = key[p] == q
        return idx[p][ok], ok

    def _add_graph(self, ga, hash_to_vid=None):
        """Append the vertices/edges of one build (rows C2b): new hashes become new vertex ids, edges are
        appended in dict order after the existing ones."""
        nv0 = self.v_hash.size
        if nv0 == 0:
            local_to_global = np.arange(ga.v_hash.size, dtype=np.int64)
            # the first build's arrays become the engine's state (graph_fn results are not used by anyone else)
            self.v_hash = np.ascontiguousarray(ga.v_hash, dtype=np.uint64)
            self.v_alive = np.ones(ga.v_hash.size, bool)
            self.v_rec = np.ascontiguousarray(ga.occ_rec, dtype=np.int64)
            self.v_pos = np.ascontiguousarray(ga.occ_pos, dtype=np.int64)
            if ga.v_hash.size > 1 and not (ga.v_hash[1:] > ga.v_hash[:-1]).all():
                order = np.argsort(ga.v_hash, kind="stable")
                self._hs, self._hid = self.v_hash[order], order
            else:                                              # (neither array is ever written in place)
                self._hs, self._hid = self.v_hash, local_to_global
        else:
            # per hash, the live vertex carrying it (a deleted vertex may share its hash with a re-created one)
            hs, hid = self._live_index()
            if hs.size:
                p = np.minimum(np.searchsorted(hs, ga.v_hash), hs.size - 1)
                hit = hs[p] == ga.v_hash
                local_to_global = np.where(hit, hid[p], -1)
            else:
                local_to_global = np.full(ga.v_hash.size, -1, np.int64)
            new = np.flatnonzero(local_to_global < 0)
            local_to_global[new] = nv0 + np.arange(new.size)
            self.v_hash = np.concatenate((self.v_hash, ga.v_hash[new]))
            self.v_alive = np.concatenate((self.v_alive, np.ones(new.size, bool)))
            self.v_rec = np.concatenate((self.v_rec, ga.occ_rec[:, new].astype(np.int64)), axis=1)
            self.v_pos = np.concatenate((self.v_pos, ga.occ_pos[:, new].astype(np.int64)), axis=1)
            # S:282-290: positions of every hash that survived the filters are overwritten
            self.v_rec[:, local_to_global] = ga.occ_rec
            self.v_pos[:, local_to_global] = ga.occ_pos
            # the new vertices join the index (ga.v_hash is ascending, so is its subset)
            nh = ga.v_hash[new]
            if nh.size > 1 and not (nh[1:] > nh[:-1]).all():
                o = np.argsort(nh, kind="stable")
                nh, new_ids = nh[o], (nv0 + np.arange(new.size))[o]
            else:
                new_ids = nv0 + np.arange(new.size)
            at = np.searchsorted(hs, nh)
            self._hs, self._hid = np.insert(hs, at, nh), np.insert(hid, at, new_ids)
        if ga.dict_ordered and nv0 == 0:                       # first build: local ids are the global ids
            eu, ev, ew = ga.e_u.astype(np.int64), ga.e_v.astype(np.int64), ga.e_w.astype(np.int64)
        elif ga.dict_ordered:
            eu, ev, ew = local_to_global[ga.e_u], local_to_global[ga.e_v], ga.e_w.astype(np.int64)
        else:
            order = dict_order(ga.e_u, ga.e_first, ga.v_hash.size)
            eu, ev, ew = local_to_global[ga.e_u[order]], local_to_global[ga.e_v[order]], ga.e_w[order].astype(np.int64)
        if nv0 and eu.size:
            # an edge that already exists keeps its slot and takes the new weight (never seen in practice); only an
            # edge between two vertices that existed before this build can be one
            old = np.flatnonzero((eu < nv0) & (ev < nv0))
            if old.size:
                hit, ok = self._find_edges(eu[old], ev[old])
                if hit.size:
                    dup = np.zeros(eu.size, bool)
                    dup[old[ok]] = True
                    self.e_w[hit] = ew[dup]
                    eu, ev, ew = eu[~dup], ev[~dup], ew[~dup]
        if self.e_u.size == 0:
            self.e_u, self.e_v, self.e_w = eu, ev, ew
        else:
            self.e_u = np.concatenate((self.e_u, eu))
            self.e_v = np.concatenate((self.e_v, ev))
            self.e_w = np.concatenate((self.e_w, ew))
        self.e_alive = np.concatenate((self.e_alive, np.ones(eu.size, bool)))
        return local_to_global

    # ------------------------------------------------------------------ C3: bubble removal (S:548-590)
    def _simplify(self, apply_deletions):
        wmax = self.G                                      # sum of the weights, all 1 (S:32, S:571)
        is3 = self._degrees() == 3                         # one byte per vertex: the edge-wise gathers stay in cache
        cand = np.flatnonzero(self.e_alive & is3[self.e_u] & is3[self.e_v])
        if cand.size == 0:
            return
        cv = np.unique(np.concatenate((self.e_u[cand], self.e_v[cand])))
        is_cv = np.zeros(self.v_hash.size, bool)
        is_cv[cv] = True
        inc = np.flatnonzero(self.e_alive & (is_cv[self.e_u] | is_cv[self.e_v]))
        adj = {}
        for e in inc.tolist():
            u, v = int(self.e_u[e]), int(self.e_v[e])
            adj.setdefault(u, {})[v] = e
            adj.setdefault(v, {})[u] = e
        doomed = []
        for e in cand.tolist():                            # ascending edge index = reference edge order
            s, t = int(self.e_u[e]), int(self.e_v[e])
            if [int(self.e_w[x]) for x in adj[s].values()].count(wmax) != 1:
                continue
            if [int(self.e_w[x]) for x in adj[t].values()].count(wmax) != 1:
                continue
            # neighbours of s other than t that also neighbour t: need t's full adjacency, which `adj`
            # holds because t is a candidate vertex too
            common = [u for u in adj[s] if u != t and u in adj[t]]
            if len(common) == 1:
                doomed.append(common[0])
                self.stats["bubbles"] += 1
                self.e_w[e] = wmax
        if apply_deletions:
            self._delete_vertices(doomed)

    # ------------------------------------------------------------------ C5/C6/C7: paths -> blocks
    # Paths are handled as one concatenated vertex array + offsets; every rule below is a segment operation.
    def _paths(self):
        m = self.e_alive
        # start at the end with the smaller position in the reference assembly (ntJoin's
        # determine_source_vertex; two vertices never share a position in one assembly)
        ref_pos = self.v_pos[self.ref]
        if getattr(self.walk_fn, "oriented", False):           # nts_walk_paths: mask and orientation handled natively
            off, verts = self.walk_fn(self.v_hash.size, self.e_u, self.e_v, e_alive=m, key=ref_pos)
            return verts, off
        off, verts = self.walk_fn(self.v_hash.size, self.e_u[m], self.e_v[m])
        if off.size < 2:
            return verts, off
        flip = ref_pos[verts[off[1:] - 1]] < ref_pos[verts[off[:-1]]]
        if flip.any():
            seg = np.repeat(np.arange(off.size - 1), np.diff(off))
            j = np.arange(verts.size)
            verts = verts[np.where(flip[seg], off[seg] + off[seg + 1] - 1 - j, j)]
        return verts, off

    def _orient_codes(self, n_up, n_d):
        """synteny_block.py:48-65 from the number of rising steps n_up among the n_d steps of a run:
        0 '+', 1 '-', 2 '?' (mixed, below the threshold self.m)."""
        with np.errstate(divide="ignore", invalid="ignore"):
            pos_perc = n_up / n_d.astype(np.float64) * 100
        neg_perc = 100 - pos_perc
        code = np.full(n_up.size, 2, np.int8)
        code[neg_perc >= self.m] = 1
        code[pos_perc >= self.m] = 0
        code[n_up == 0] = 1
        code[n_up == n_d] = 0                                  # includes single-vertex runs
        return code

    def _blocks_of_paths(self, paths):
        """C6-C8.  The per-vertex work (contig changes, rising steps, gap spreads) is one threaded pass in
        nts_path_scan; the rules are applied to its per-path results."""
        verts, off = paths
        if off.size < 2:
            return []
        start, n_up, over = self.scan_fn(self.v_rec, self.v_pos, off, verts, self.bp)
        end = off[1:]
        codes = [self._orient_codes(n_up[a], end - start - 1) for a in range(self.G)]
        good = np.ones(start.size, bool)
        for c in codes:
            good &= c != 2
        bad = np.flatnonzero(~good)
        if bad.size:                                           # S:499-505: blocks without an orientation are dropped
            self._delete_vertices(np.concatenate([verts[start[i]:end[i]] for i in bad.tolist()]))
            self.stats["unoriented"] += int(bad.size)
            for i in bad.tolist():
                over[start[i]:end[i]] = False
        # C8: indel split (S:364-409): cut between verts[c] and verts[c + 1]
        cuts = np.flatnonzero(over)
        if cuts.size:
            self.stats["indel_cuts"] += int(cuts.size)
            dead, _ = self._find_edges(verts[cuts], verts[cuts + 1])
            self.e_alive[dead] = False
        owner = np.searchsorted(end, cuts, side="right")       # path of each cut
        out = []
        sym = "+-?"
        first = verts[start]
        recs = [self.v_rec[a][first] for a in range(self.G)]
        for i in np.flatnonzero(good).tolist():
            rec = [int(r[i]) for r in recs]
            ori = [sym[c[i]] for c in codes]
            lo, hi = np.searchsorted(owner, [i, i + 1]) if cuts.size else (0, 0)
            if lo == hi:
                out.append(Block(verts[start[i]:end[i]], rec, ori))
                continue
            bounds = [int(start[i])] + (cuts[lo:hi] + 1).tolist() + [int(end[i])]
            for x, y in zip(bounds[:-1], bounds[1:]):
                out.append(Block(verts[x:y], list(rec), list(ori)))   # contig / orientation inherited (S:383)
        return out

    # ------------------------------------------------------------------ C9 (S:411-426)
    def _drop_small(self, blocks, min_mx):
        keep, drop = [], []
        for b in blocks:
            (keep if b.vids.size >= min_mx else drop).append(b)
        if drop:
            self._delete_vertices(np.concatenate([b.vids for b in drop]))
            self.stats["small_blocks"] += len(drop)
        return keep

    # ------------------------------------------------------------------ C10: geometry, order, text
    def _finish(self, b):
        if not b.first_pos:
            b.first_pos = [int(self.v_pos[a][b.vids[0]]) for a in range(self.G)]
            b.last_pos = [int(self.v_pos[a][b.vids[-1]]) for a in range(self.G)]
            b.n_mx = int(b.vids.size)
        return b

    def _start(self, b, a):
        return min(b.first_pos[a], b.last_pos[a])

    def _end(self, b, a):
        return max(b.first_pos[a], b.last_pos[a]) + self.k

    def _finish_all(self, blocks):
        "first/last positions of every block that does not have them yet, gathered per assembly in one indexing each"
        todo = [b for b in blocks if not b.first_pos]
        if not todo:
            return
        first = np.array([b.vids[0] for b in todo], np.int64)
        last = np.array([b.vids[-1] for b in todo], np.int64)
        fp = [self.v_pos[a][first].tolist() for a in range(self.G)]
        lp = [self.v_pos[a][last].tolist() for a in range(self.G)]
        for i, b in enumerate(todo):
            b.first_pos = [fp[a][i] for a in range(self.G)]
            b.last_pos = [lp[a][i] for a in range(self.G)]
            b.n_mx = int(b.vids.size)

    def _long_enough(self, b):
        need = self.z - self.k                                 # end - start = |first - last| + k
        for f, l in zip(b.first_pos, b.last_pos):
            if abs(f - l) < need:
                return False
        return True

    def _sorted(self, blocks):
        self._finish_all(blocks)
        ref, names = self.ref, self.contigs[self.ref]
        return sorted(blocks, key=lambda b: (names[b.rec[ref]], min(b.first_pos[ref], b.last_pos[ref])))

    def _text(self, b, num, verbose):
        rows = []
        if self._asm_names is None:                            # assembly names as printed: the TSV name without its suffix
            self._asm_names = []
            for f in self.files:
                mt = MX_SUFFIX.search(f)
                self._asm_names.append(mt.group(1) if mt else f)
        for a in self.out_order:
            name = self._asm_names[a]
            row = f"{num}\t{name}\t{self.contigs[a][b.rec[a]]}\t{self._start(b, a)}\t{self._end(b, a)}\t{b.ori[a]}\t{b.n_mx}"
            if verbose:
                row = f"{row.strip()}\t{b.reason}"
            rows.append(row + "\n")
        return "".join(rows)

    def _emit(self, name, blocks, verbose=False):
        rows, num = [], 0
        for b in blocks:
            if self._long_enough(b):
                rows.append(self._text(b, num, verbose))
                num += 1
        text = "".join(rows)
        self.outputs[name] = text
        with open(name, "w", encoding="utf-8") as fh:
            fh.write(text)

    # ------------------------------------------------------------------ C12: collinear merge (S:428-472)
    def _merge(self, blocks):
        out = []
        cur = blocks[0]
        for b in blocks[1:]:
            same_ori = all(cur.ori[a] == b.ori[a] for a in range(self.G))
            same_ctg = all(cur.rec[a] == b.rec[a] for a in range(self.G))
            diffs = []
            for a in range(self.G):
                if cur.ori[a] == "-" and b.ori[a] == "-":
                    diffs.append(self._start(cur, a) - self._end(b, a))
                else:
                    diffs.append(self._start(b, a) - self._end(cur, a))
            spread = max(diffs) - min(diffs)
            if (not same_ori) or (not same_ctg) or spread > self.bp - self.k or max(diffs) >= self.collinear_merge:
                if not same_ctg:
                    b.reason = "id_change"
                elif not same_ori:
                    b.reason = "ori_change"
                elif any(d < 0 for d in diffs):
                    b.reason = "inconsistent_order"
                elif spread > self.bp - self.k:
                    b.reason = "indel"
                elif max(diffs) >= self.collinear_merge:
                    b.reason = "merge"
                out.append(cur)
                cur = b
            else:
                cur.last_pos = list(b.last_pos)            # minimizers.extend(): only ends and count matter
                cur.n_mx += b.n_mx
                self.stats["merged"] += 1
        out.append(cur)
        return out

    # ------------------------------------------------------------------ --dev: overlap self-check (S:234-253)
    def _warn_overlaps(self, rec, start, end):
        """check_non_overlapping over the final blocks (all of them pass the length filter by then): rec / start / end are
        [assembly][block] in final order; a block warns -- once per assembly -- when its extent overlaps an EARLIER block's
        extent in the same assembly and contig by at least z.  Per assembly the blocks are swept in (contig, start) order with
        the few extents still open; the warnings come out in the reference's order (block, then assembly)."""
        hits = []
        for a in range(self.G):
            r, s0, e0 = np.asarray(rec[a]), np.asarray(start[a], np.int64), np.asarray(end[a], np.int64)
            order = np.lexsort((s0, r)).tolist()
            rl, sl, el = r.tolist(), s0.tolist(), e0.tolist()
            warned, active, cur = set(), [], None
            for i in order:
                if rl[i] != cur:
                    cur, active = rl[i], []
                active = [j for j in active if el[j] > sl[i]]
                for j in active:
                    if min(el[j], el[i]) - max(sl[j], sl[i]) >= self.z:
                        warned.add(max(i, j))                # the later block in the final order is the one being checked
                active.append(i)
            hits += [(b, a) for b in warned]
        for b, a in sorted(hits):
            print("WARNING: detected overlapping segments for this block:", self.files[a], self.contigs[a][int(rec[a][b])],
                  int(start[a][b]), int(end[a][b]), "\n", file=sys.stderr, flush=True)

    # ------------------------------------------------------------------ B5 + C11: refinement inputs
    def _mask_intervals(self, blocks, w):
        masks = [[] for _ in range(self.G)]
        lim = max(2 * w, w + self.k + 1)
        self._finish_all(blocks)
        for b in blocks:
            for a in range(self.G):
                s, e = self._start(b, a), self._end(b, a)
                if e - s > lim:
                    s2, e2 = s + (w + self.k), e - (w + self.k)
                    if e2 > s2:
                        masks[a].append((b.rec[a], s2, e2))
        return masks

    def _new_round_graph(self, blocks, new_w, prev_w):
        if not blocks:
            # S:134-146 masks, and S:150-192 sketches again, the assemblies the blocks name: a round that starts without a block reads
            # no assembly at all and adds nothing to the graph (the run then goes on over the graph as the last round left it)
            return np.zeros(self.v_hash.size, bool)
        masks = self._mask_intervals(blocks, prev_w)
        nv = self.v_hash.size
        internal = np.zeros(nv, bool)
        terminal = np.zeros(nv, bool)
        first = np.array([b.vids[0] for b in blocks], np.int64)
        last = np.array([b.vids[-1] for b in blocks], np.int64)
        terminal[first] = True
        terminal[last] = True
        if blocks:
            inner = np.concatenate([b.vids[1:-1] for b in blocks])
            internal[inner] = True
        # block interiors [min+1, max) per assembly, per contig (S:194-203): per assembly the records and the intervals
        # of all blocks as arrays (tens of thousands of blocks in a fragmented assembly)
        spans = []
        for a in range(self.G):
            p0, p1 = self.v_pos[a][first], self.v_pos[a][last]
            lo, hi = np.minimum(p0, p1), np.maximum(p0, p1)
            ok = hi - lo >= 2
            recs = np.array([b.rec[a] for b in blocks], np.int64)
            spans.append((recs[ok], lo[ok] + 1, hi[ok]))
        hs, hid = self._live_index()
        uh = hs[internal[hid]]                             # hashes of the live internal vertices, ascending
        lists, keeps, list_ids = [], [], []
        # all assemblies of the round at once when the caller can (one batch on the GPU, one exchange across GPUs)
        sketched = None
        if getattr(self.sketch_fn, "all_at_once", None) is not None:
            by_input = {self.input_order[a]: masks[a] for a in range(self.G)}
            res = self.sketch_fn.all_at_once(by_input, new_w)
            sketched = [res[self.input_order[a]] for a in range(self.G)]
        for a in range(self.G):
            h1, rec, pos = sketched[a] if sketched is not None else self.sketch_fn(self.input_order[a], masks[a], new_w)
            h1 = np.asarray(h1, np.uint64)
            rec = np.asarray(rec, np.int64)
            pos = np.asarray(pos, np.int64)
            # C1: hashes seen once in this (masked) assembly
            # (one sort of the list serves both this and the look-up below: sorted queries walk `uh` in order)
            order = np.argsort(h1, kind="stable")
            sh = h1[order]
            first = np.ones(sh.size, bool)
            first[1:] = sh[1:] != sh[:-1]
            run = np.cumsum(first) - 1
            uniq = np.empty(h1.size, bool)
            uniq[order] = (np.bincount(run) == 1)[run] if sh.size else np.zeros(0, bool)
            # is the hash an internal minimizer of a block?  (S:274: `mx not in black_list`)
            is_internal = np.zeros(h1.size, bool)
            if uh.size and sh.size:
                p = np.minimum(np.searchsorted(uh, sh), uh.size - 1)
                is_internal[order] = uh[p] == sh
            inside = np.zeros(h1.size, bool)
            cut_before = np.zeros(h1.size, bool)
            # "does [s, e) of record r reach into a block interior of r?" for all records at once: intervals and queries
            # as composite keys record * 2^40 + position; the running maximum of the interval ends (S:194-203's merged
            # intervals) never carries over from one record to the next because a later record's keys are all larger
            OFF = np.int64(1) << 40
            ir, iv_s, iv_e = spans[a]
            have = ir.size > 0
            if have and (int(ir.max()) >= (1 << 22) or (pos.size and int(pos.max()) >= (1 << 40) - 1)):
                raise ValueError("more than 2^22 records or a record beyond 2^40 bases: composite interval keys would overflow")
            if have:
                order = np.argsort(ir * OFF + iv_s, kind="stable")
                comp_s = (ir * OFF + iv_s)[order]
                comp_mx = np.maximum.accumulate((ir * OFF + iv_e)[order])

                def overlaps(r, s0, e0):
                    base = r * OFF
                    i = np.searchsorted(comp_s, base + e0, side="left")
                    j = np.maximum(i - 1, 0)
                    return (i > 0) & (comp_s[j] >= base) & (comp_mx[j] > base + s0) & (e0 > s0)
                sel = np.flatnonzero(uniq)
                inside[sel] = overlaps(rec[sel], pos[sel], pos[sel] + 1)
            kept = uniq & ~is_internal & ~inside
            # cut a list wherever the span between two consecutive kept minimizers (of one record) crosses a block interior
            if have:
                sel = np.flatnonzero(kept)
                if sel.size > 1:
                    same = rec[sel[1:]] == rec[sel[:-1]]
                    ov = same & overlaps(rec[sel[1:]], pos[sel[:-1]], pos[sel[1:]])
                    cut_before[sel[1:][ov]] = True
            # list ids: a new list at every record change and at every cut
            new_list = np.ones(h1.size, bool)
            if h1.size:
                new_list[1:] = (rec[1:] != rec[:-1]) | cut_before[1:]
            lid = np.cumsum(new_list) - 1
            lists.append((h1, rec, pos))
            keeps.append(kept)
            list_ids.append(lid)
        ga = self.graph_fn(lists, keeps, list_ids)
        self._add_graph(ga)
        return terminal

    # ------------------------------------------------------------------ last-round erosion (S:292-362)
    def _refine_graph(self, flagged):
        if flagged[0].size == 0:
            return
        deg = self._degrees()
        m = self.e_alive
        idx = np.flatnonzero(m)
        inc = {}
        involved = set()

        def neighbours(v):
            if v not in inc:
                sel = idx[(self.e_u[idx] == v) | (self.e_v[idx] == v)]
                inc[v] = [(int(e), int(self.e_v[e]) if int(self.e_u[e]) == v else int(self.e_u[e])) for e in sel]
            return inc[v]

        def too_close(a, b):
            return bool((np.abs(self.v_pos[:, a] - self.v_pos[:, b]) < self.k).any())
        dead = set()
        for s, t in zip(flagged[0].tolist(), flagged[1].tolist()):
            if str(int(self.v_hash[s])) > str(int(self.v_hash[t])):
                s, t = t, s
            if deg[s] != 1 or deg[t] != 1:
                continue
            erode_target, cs, ct = True, s, t
            visited = {s, t}
            while too_close(cs, ct):
                v = ct if erode_target else cs
                nb = neighbours(v)
                dead.update(e for e, _ in nb)
                nxt = [u for _, u in nb if u not in visited]
                if not nxt:
                    break
                assert len(nxt) == 1
                if erode_target:
                    ct = nxt[0]
                    visited.add(ct)
                else:
                    cs = nxt[0]
                    visited.add(cs)
                erode_target = not erode_target
        if dead:
            self.e_alive[np.fromiter(dead, dtype=np.int64)] = False
            self.stats["eroded_edges"] += len(dead)
        del involved

    # ------------------------------------------------------------------ drivers (S:476-530, S:593-647)
    def _round_blocks(self):
        return self._drop_small(self._blocks_of_paths(self._paths()), 4)

    def _write_interarrivals(self, vid_lists, v_pos=None):
        """<prefix>.interarrivals.tsv (S:557-564): per block, per assembly (in the reference's assembly order), the distance between
        every two neighbouring minimizers, one per line.  The blocks come in this engine's path order, which is not the
        reference's (ntJoin's component order is not pinned): the file holds the same lines in a different block order."""
        v_pos = self.v_pos if v_pos is None else v_pos
        parts = []
        for vids in vid_lists:
            vids = np.asarray(vids, dtype=np.int64)
            for a in range(self.G):
                parts.append(np.abs(np.diff(v_pos[a][vids].astype(np.int64))))
        flat = np.concatenate(parts) if parts else np.zeros(0, np.int64)
        text = "".join(f"{int(x)}\n" for x in flat)
        name = f"{self.prefix}.interarrivals.tsv"
        self.outputs[name] = text
        with open(name, "w", encoding="utf-8") as fout:
            fout.write(text)

    def run(self, initial_lists):
        """initial_lists[i] = (h1, rec, pos) of assembly i in the caller's order."""
        if len(self.w_rounds) != len(set(self.w_rounds)):
            print("Error: duplicate values found in w_rounds!", file=sys.stderr, flush=True)
            sys.exit(1)
        lists = [initial_lists[i] for i in self.input_order]
        ga = self.graph_fn(lists, None, None)
        self._add_graph(ga)
        if self.simplify:
            self._simplify(apply_deletions=True)
        if self.n > 1:
            self.e_alive &= self.e_w >= self.n
        blocks = self._round_blocks()
        if self.interarrivals:
            self._write_interarrivals([b.vids for b in blocks])
        ordered = self._sorted(blocks)
        if not ordered:
            print("Error - no paths found. Try adjusting the specified k/w parameters.")
            sys.exit(1)
        self._emit(f"{self.prefix}.synteny_blocks.tsv", ordered)
        prev_w = self.w
        for new_w in self.w_rounds:
            self.log(f"Extending synteny blocks with w = {new_w}")
            self._new_round_graph(blocks, new_w, prev_w)
            if self.simplify:
                self._simplify(apply_deletions=False)
            last = new_w == self.w_rounds[-1]
            light = self.e_alive & (self.e_w < self.n)
            flagged = (self.e_u[light], self.e_v[light])
            if last or self.n > 1:
                self.e_alive &= ~light
            if last:
                self._refine_graph(flagged)
            blocks = self._round_blocks()
            ordered = self._sorted(blocks)
            self._emit(f"{self.prefix}.pre-collinear-merge.synteny_blocks.tsv", ordered)
            if last:
                # S:505-510 calls merge_collinear_blocks twice, unconditionally: with no block left, or none of at least z bases, the
                # reference ends in an IndexError at S:437 -- the run fails there as well (pipeline.run then removes the block tables, as
                # Snakemake removes a failed rule's outputs)
                if not ordered:
                    raise IndexError(EMPTY_FINAL.format(""))
                merged = self._merge(ordered)
                merged = [b for b in merged if self._long_enough(b)]
                if not merged:
                    raise IndexError(EMPTY_FINAL.format(" of at least z bases"))
                merged = self._merge(merged)
                if self.dev and merged:
                    self._warn_overlaps([[b.rec[a] for b in merged] for a in range(self.G)],
                                        [[self._start(b, a) for b in merged] for a in range(self.G)],
                                        [[self._end(b, a) for b in merged] for a in range(self.G)])
                self._emit(f"{self.prefix}.synteny_blocks.tsv", merged, verbose=True)
            prev_w = new_w
        return self.outputs
