"""Deterministic synthetic genomes for parity tests and bench.py (SURVEY.md 8(d) "Synthetic inputs").

ancestor: i.i.d. uniform A/C/G/T split into contigs; genome j = ancestor with independent per-base
substitutions at rate p/2 (pairwise divergence ~ p) plus a small fixed set of structural events
(inversions, inter-contig translocations, insertions/deletions) so that `ori_change`, `id_change`
and `indel` block breaks are exercised; optional N-runs and soft-masked (lower-case) stretches.
"""
import numpy as np

BASE_SEED = 20240207
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    _COMP[_a] = _b


def random_dna(n, rng):
    return _ACGT[rng.integers(0, 4, size=n, dtype=np.uint8)]


def revcomp(a):
    return _COMP[a[::-1]]


def make_ancestor(total_bp, n_contigs, seed=BASE_SEED):
    rng = np.random.Generator(np.random.PCG64(seed))
    per = total_bp // n_contigs
    return [random_dna(per, rng) for _ in range(n_contigs)]


def _micro_events(contigs, rng, n_events):
    """Small rearrangements (0.3-3 kbp segments moved / copied / inverted a few tens of kbp away): they
    create bubbles in the minimizer graph, short blocks, and nearby collinear blocks that merge."""
    out = []
    for g in contigs:
        g = g.copy()
        for _ in range(n_events):
            ln = int(rng.integers(300, 3000))
            if g.size < 20 * ln:
                break
            st = int(rng.integers(ln, g.size - 2 * ln))
            kind = rng.random()
            seg = g[st:st + ln].copy()
            if kind < 0.4:      # move
                rest = np.concatenate([g[:st], g[st + ln:]])
                dst = int(np.clip(st + rng.integers(-60000, 60000), ln, rest.size - ln))
                g = np.concatenate([rest[:dst], seg, rest[dst:]])
            elif kind < 0.7:    # copy
                dst = int(np.clip(st + rng.integers(-60000, 60000), ln, g.size - ln))
                g = np.concatenate([g[:dst], seg, g[dst:]])
            else:               # invert in place
                g[st:st + ln] = revcomp(seg)
        out.append(g)
    return out


def derive_genome(ancestor, divergence, j, seed=BASE_SEED, structural=True, n_runs=False,
                  soft_mask=False, micro=0):
    """Genome j of a family.  `divergence` is the pairwise fraction (0.01 for 1 %)."""
    rng = np.random.Generator(np.random.PCG64(seed + 1 + j))
    contigs = []
    for c in ancestor:
        g = c.copy()
        if divergence > 0:
            hit = rng.random(g.size) < (divergence / 2.0)
            idx = np.nonzero(hit)[0]
            # substitute with one of the three other bases
            code = np.searchsorted(_ACGT, g[idx])
            g[idx] = _ACGT[(code + rng.integers(1, 4, size=idx.size)) % 4]
        contigs.append(g)
    if structural and j > 0:
        contigs = _structural_events(contigs, rng)
    if micro and j > 0:
        contigs = _micro_events(contigs, rng, micro)
    if n_runs:
        for g in contigs:
            n_ev = max(1, g.size // 200000)
            for _ in range(n_ev):
                ln = int(rng.integers(1, max(2, min(50000, g.size // 50))))
                st = int(rng.integers(0, max(1, g.size - ln)))
                g[st:st + ln] = ord("N")
    if soft_mask:
        for g in contigs:
            n_ev = max(1, g.size // 100000)
            for _ in range(n_ev):
                ln = int(rng.integers(10, max(11, min(20000, g.size // 40))))
                st = int(rng.integers(0, max(1, g.size - ln)))
                seg = g[st:st + ln]
                up = seg != ord("N")
                seg[up] = seg[up] | 0x20
    return contigs


def _structural_events(contigs, rng):
    contigs = [c for c in contigs]
    total = sum(c.size for c in contigs)
    # inversions: 2 per genome, 0.5-2 % of a contig each
    for _ in range(2):
        ci = int(rng.integers(0, len(contigs)))
        g = contigs[ci]
        ln = int(g.size * rng.uniform(0.005, 0.02))
        if ln < 50 or g.size < 4 * ln:
            continue
        st = int(rng.integers(ln, g.size - 2 * ln))
        g[st:st + ln] = revcomp(g[st:st + ln].copy())
    # one translocation between two contigs (if there are at least two)
    if len(contigs) >= 2:
        a, b = rng.choice(len(contigs), size=2, replace=False)
        ga, gb = contigs[int(a)], contigs[int(b)]
        ln = int(min(ga.size, gb.size) * rng.uniform(0.01, 0.03))
        if ln >= 50 and ga.size > 4 * ln and gb.size > 4 * ln:
            st = int(rng.integers(ln, ga.size - 2 * ln))
            seg = ga[st:st + ln].copy()
            contigs[int(a)] = np.concatenate([ga[:st], ga[st + ln:]])
            ins = int(rng.integers(ln, gb.size - ln))
            contigs[int(b)] = np.concatenate([gb[:ins], seg, gb[ins:]])
    # indels: 4 per genome, up to ~0.2 % of the genome (>= the small-divergence indel thresholds
    # at test sizes would be unrealistic, so sizes scale with the genome)
    for _ in range(4):
        ci = int(rng.integers(0, len(contigs)))
        g = contigs[ci]
        ln = int(max(30, total * rng.uniform(0.0002, 0.002)))
        if g.size < 6 * ln:
            continue
        st = int(rng.integers(ln, g.size - 2 * ln))
        if rng.random() < 0.5:
            contigs[ci] = np.concatenate([g[:st], g[st + ln:]])
        else:
            contigs[ci] = np.concatenate([g[:st], random_dna(ln, rng), g[st:]])
    return contigs


def write_fasta(path, contigs, names=None, line_width=0):
    names = names or [f"chr{i + 1}" for i in range(len(contigs))]
    with open(path, "wb") as fh:
        for name, g in zip(names, contigs):
            fh.write(b">" + name.encode() + b"\n")
            if line_width and line_width > 0:
                for i in range(0, g.size, line_width):
                    fh.write(g[i:i + line_width].tobytes() + b"\n")
            else:
                fh.write(g.tobytes() + b"\n")
    return path


def make_family(outdir, n_genomes, total_bp, n_contigs, divergence, seed=BASE_SEED, prefix="syn",
                structural=True, n_runs=False, soft_mask=False, line_width=0, micro=0):
    """Write `n_genomes` FASTA files; returns their paths."""
    import os
    anc = make_ancestor(total_bp, n_contigs, seed)
    paths = []
    for j in range(n_genomes):
        g = derive_genome(anc, divergence, j, seed, structural, n_runs, soft_mask, micro)
        p = os.path.join(outdir, f"{prefix}{j}.fa")
        write_fasta(p, g, line_width=line_width)
        paths.append(p)
    return paths


# ---- families generated in HBM (bench.py, scale tests): the plan of genome j's structural events ------------------------------
# "a fixed small set of structural events per genome (e.g. 5 inversions of 1-5 Mbp, 2 inter-contig
# translocations, 20 indels of 1-60 kbp) so that block breaks (ori_change, id_change, indel) are exercised", and a variant with
# N runs (0.5 % of the bases in runs of 100-50,000).  The genome is described as a tiling of pieces of the ancestor
# (include/ntsynt_hip.h nts_synth_piece); the bases themselves are generated on the device (nts_genome_synth_plan).
PIECE_DTYPE = np.dtype([("dst", np.uint64), ("len", np.uint64), ("src", np.uint64), ("flags", np.uint32), ("reserved", np.uint32)])
REVCOMP, NOVEL, NRUN = 1, 2, 4


class _PieceTable:
    "one contig as a list of [src, len, flags, family]"

    def __init__(self, src, length):
        self.p = [[int(src), int(length), 0, 0]]

    def size(self):
        return sum(x[1] for x in self.p)

    def split(self, pos):
        "piece boundary at contig coordinate pos; returns the index of the piece that starts there"
        at = 0
        for i, (src, ln, fl, fam) in enumerate(self.p):
            if pos == at:
                return i
            if pos < at + ln:
                cut = pos - at
                if fl & REVCOMP:        # a reversed piece reads its source backwards: the head of the piece is the tail of the source
                    first, second = [src + ln - cut, cut, fl, fam], [src, ln - cut, fl, fam]
                else:
                    first, second = [src, cut, fl, fam], [src + cut, ln - cut, fl, fam]
                self.p[i:i + 1] = [first, second]
                return i + 1
            at += ln
        assert pos == at
        return len(self.p)

    def cut(self, a, ln):
        i0 = self.split(a)
        i1 = self.split(a + ln)
        out = self.p[i0:i1]
        del self.p[i0:i1]
        return out

    def copy(self, a, ln):
        i0 = self.split(a)
        i1 = self.split(a + ln)
        return [list(x) for x in self.p[i0:i1]]

    def insert(self, a, pieces):
        i = self.split(a)
        self.p[i:i] = pieces

    def invert(self, a, ln):
        seg = self.cut(a, ln)
        self.insert(a, _reversed_pieces(seg))


def _reversed_pieces(seg):
    return [[s_, n, f ^ REVCOMP, fam] for s_, n, f, fam in reversed(seg)]


def _structural_events_on_tables(tabs, rng, scale, inversions, translocations, indels, indel_bp, micro, micro_bp, micro_shift, small_indels):
    "genome j's own events on its piece tables; returns the length of the genome's novel stream used"
    n_contigs = len(tabs)
    novel = 0
    lo_i, hi_i = indel_bp or (max(30, int(1000 * scale)), max(120, int(60000 * scale)))
    for _ in range(inversions):
        t = tabs[int(rng.integers(0, n_contigs))]
        ln = int(rng.uniform(1e6, 5e6) * scale)
        if ln < 60 or t.size() < 6 * ln:
            continue
        t.invert(int(rng.integers(ln, t.size() - 2 * ln)), ln)
    for _ in range(translocations if n_contigs >= 2 else 0):
        a, b = (int(x) for x in rng.choice(n_contigs, size=2, replace=False))
        ln = int(rng.uniform(0.5e6, 2e6) * scale)
        if ln < 60 or tabs[a].size() < 6 * ln or tabs[b].size() < 6 * ln:
            continue
        seg = tabs[a].cut(int(rng.integers(ln, tabs[a].size() - 2 * ln)), ln)
        tabs[b].insert(int(rng.integers(ln, tabs[b].size() - ln)), seg)
    for _ in range(indels):
        t = tabs[int(rng.integers(0, n_contigs))]
        ln = int(rng.integers(lo_i, hi_i + 1))
        if t.size() < 8 * ln:
            continue
        at = int(rng.integers(ln, t.size() - 2 * ln))
        if rng.random() < 0.5:
            t.cut(at, ln)
        else:
            t.insert(at, [[novel, ln, NOVEL, 0]])
            novel += ln
    lo_m, hi_m = micro_bp or (max(40, int(2000 * scale)), max(80, int(20000 * scale)))
    shift = micro_shift or max(200, int(80000 * scale))
    for _ in range(micro):
        t = tabs[int(rng.integers(0, n_contigs))]
        ln = int(rng.integers(lo_m, hi_m + 1))
        if t.size() < 4 * (ln + shift):
            continue
        st = int(rng.integers(ln + shift, t.size() - 2 * (ln + shift)))
        kind = rng.random()
        if kind < 0.4:                          # move
            seg = t.cut(st, ln)
            t.insert(st + int(rng.integers(-shift, shift + 1)), seg)
        elif kind < 0.7:                        # copy
            seg = t.copy(st, ln)
            t.insert(st + int(rng.integers(-shift, shift + 1)), seg)
        else:                                   # invert in place
            t.invert(st, ln)
    for _ in range(small_indels):
        t = tabs[int(rng.integers(0, n_contigs))]
        ln = int(rng.integers(1, 51))
        at = int(rng.integers(1000, t.size() - 1000)) if t.size() > 4000 else 0
        if not at:
            continue
        if rng.random() < 0.5:
            t.cut(at, ln)
        else:
            t.insert(at, [[novel, ln, NOVEL, 0]])
            novel += ln
    return novel


def _pieces_array(rows):
    pieces = np.zeros(len(rows), dtype=PIECE_DTYPE)
    pieces["src"] = [r[0] for r in rows]
    pieces["len"] = [r[1] for r in rows]
    pieces["flags"] = [r[2] for r in rows]
    pieces["reserved"] = [r[3] for r in rows]
    pieces["dst"] = np.concatenate(([0], np.cumsum(pieces["len"][:-1]))).astype(np.uint64)
    return pieces


def structural_plan(n_contigs, contig_bp, j, seed=BASE_SEED, inversions=5, translocations=2, indels=20, n_runs=False,
                    indel_bp=None, micro=60, micro_bp=None, micro_shift=None, small_indels=200):
    """(record lengths, pieces) of genome j of a family whose ancestor is n_contigs x contig_bp.  Sizes follow SURVEY.md 8(d) at
    human scale (contigs of 125 Mbp) and shrink with the contigs below that; deterministic in (seed, j).

    On top of SURVEY's list: `micro` small rearrangements (a segment of micro_bp bases moved or copied up to micro_shift bases
    away, or inverted in place -- bubbles, short blocks, and neighbouring collinear blocks for the merge rule) and
    `small_indels` of 1-50 bases (neighbouring minimizers that are adjacent in some assemblies only: light edges for the last
    round's erosion)."""
    rng = np.random.Generator(np.random.PCG64(seed + 7919 * (j + 1)))
    scale = min(1.0, contig_bp / 125e6)
    tabs = [_PieceTable(c * contig_bp, contig_bp) for c in range(n_contigs)]
    _structural_events_on_tables(tabs, rng, scale, inversions, translocations, indels, indel_bp, micro, micro_bp, micro_shift, small_indels)
    if n_runs:                                  # 0.5 % of the bases in runs of 100-50,000 (scaled)
        for t in tabs:
            budget = int(0.005 * t.size())
            while budget > 0:
                ln = int(min(budget, rng.integers(max(2, int(100 * scale)), max(3, int(50000 * scale)) + 1)))
                if t.size() < 4 * ln:
                    break
                at = int(rng.integers(0, t.size() - ln))
                t.cut(at, ln)
                t.insert(at, [[0, ln, NRUN, 0]])
                budget -= ln
    rec_len = np.array([t.size() for t in tabs], dtype=np.uint64)
    return rec_len, _pieces_array([x for t in tabs for x in t.p])


# ---- the assembly-like family (BASELINE config 5 is quoted on human / chimp / bonobo assemblies, reference README.md:157; they are not
# in the container) -------------------------------------------------------------------------------------------------------------------
# What an i.i.d. ancestor lacks and a mammalian assembly has: interspersed repeat families in 10^5-10^6 copies (young copies share
# k-mers: minimizers that occur twice within an assembly and are dropped by row C1, Bloom buckets that overflow, tiles that list
# many equal candidates), satellite arrays, segmental duplications, thousands of scaffolds with a long tail shorter than a window,
# N gaps at the joins, soft-masked (lower-case) repeats.  The repeat families are a function of the ancestor coordinate evaluated on
# the device (include/ntsynt_hip.h nts_synth_repeats); satellite arrays, duplications, scaffolds and gaps are pieces of the plan.
REPEATS = {
    # SINE-like: 300-base elements, one per 1024-base cell in a third of the cells (~10 % of the sequence), four families
    "sine_cell_log2": 10, "sine_len": 300, "sine_prob_256": 85, "sine_families": 4,
    # LINE-like: the last 900-6000 bases of a 6 kbp consensus per 16384-base cell in 3/4 of the cells (~16 %), two families
    "line_cell_log2": 14, "line_len": 6000, "line_min_len": 900, "line_prob_256": 192, "line_families": 2,
    # a copy's divergence from its consensus: sixteen levels from 1 % to 20 %
    "div_min_1024": 10, "div_max_1024": 205,
    # satellites: 171-base unit, 2 % between units
    "sat_unit": 171, "sat_div_1024": 20,
}
TANDEM = 8


def _slice_rows(rows, starts, a, b):
    "the rows (pieces) of a contig that cover [a, b) of its coordinates; starts = cumulative starts of rows (+ total at the end)"
    import bisect
    out = []
    i = bisect.bisect_right(starts, a) - 1
    while i < len(rows) and starts[i] < b:
        src, ln, fl, fam = rows[i]
        lo, hi = max(a, starts[i]), min(b, starts[i] + ln)
        if hi > lo:
            head, n = lo - starts[i], hi - lo
            if fl & (NRUN | NOVEL) == NRUN:
                out.append([0, n, fl, fam])
            elif fl & REVCOMP:
                out.append([src + ln - head - n, n, fl, fam])
            else:
                out.append([src + head, n, fl, fam])
        i += 1
    return out


def realistic_plan(n_chrom, chrom_bp, j, seed=BASE_SEED, n_scaffolds=None, n_tail=None, n_gaps=None, segdups=None, sat_families=3,
                   tail_bp=(200, 1500), sat_scale=None, **events):
    """(record lengths, pieces, record names) of genome j of an assembly-like family: ancestor = n_chrom chromosomes of chrom_bp bases
    with satellite arrays and segmental duplications (the same in every genome: seeded by `seed` alone), genome j = the ancestor with
    structural_plan's events of its own, cut into n_scaffolds scaffolds plus n_tail short ones (tail_bp bases: most below w + k, so
    they give no minimizer -- the long tail of unplaced scaffolds), with n_gaps N gaps inside the scaffolds.  Use with
    Genome.synth_plan(..., rep=REPEATS)."""
    total = n_chrom * chrom_bp
    scale = min(1.0, chrom_bp / 125e6)
    rng_a = np.random.Generator(np.random.PCG64(seed + 104729))           # the ancestor's events: shared by the family
    tabs = [_PieceTable(c * chrom_bp, chrom_bp) for c in range(n_chrom)]
    sat_cursor = [0] * sat_families
    for t in tabs:                                                       # a centromere-like array + a few small ones per chromosome
        # (sat_scale: tests keep the arrays near their human size in a small genome, so that the copies of a unit overflow a bucket)
        sizes = [int(rng_a.uniform(0.5e6, 3e6) * (scale if sat_scale is None else sat_scale))] + \
                [int(rng_a.uniform(5e3, 5e4) * max(scale, 0.2)) for _ in range(3)]
        for ln in sizes:
            ln = max(ln, 20 * REPEATS["sat_unit"])
            fam = int(rng_a.integers(0, sat_families))
            at = int(rng_a.integers(1000, t.size() - 1000))
            t.insert(at, [[sat_cursor[fam], ln, TANDEM, fam]])
            sat_cursor[fam] += ln
    n_sd = segdups if segdups is not None else max(12, int(round(300 * total / 3e9)))
    sd_scale = max(scale, 0.1)
    for _ in range(n_sd):                                                # segmental duplications: 10-100 kbp, a third inverted
        a = int(rng_a.integers(0, n_chrom))
        ln = int(rng_a.uniform(1e4, 1e5) * sd_scale)
        if tabs[a].size() < 8 * ln:
            continue
        st = int(rng_a.integers(ln, tabs[a].size() - 2 * ln))
        seg = tabs[a].copy(st, ln)
        if rng_a.random() < 0.33:
            seg = _reversed_pieces(seg)
        if rng_a.random() < 0.7:                                         # nearby on the same chromosome
            b = a
            dst = int(np.clip(st + rng_a.integers(-int(5e6 * sd_scale), int(5e6 * sd_scale) + 1), ln, tabs[b].size() - ln))
        else:
            b = int(rng_a.integers(0, n_chrom))
            dst = int(rng_a.integers(ln, tabs[b].size() - ln))
        tabs[b].insert(dst, seg)
    # genome j's own events
    rng = np.random.Generator(np.random.PCG64(seed + 7919 * (j + 1)))
    ev = dict(inversions=5, translocations=2, indels=20, indel_bp=None, micro=60, micro_bp=None, micro_shift=None, small_indels=200)
    ev.update(events)
    _structural_events_on_tables(tabs, rng, scale, **ev)
    # scaffolds, the tail of short ones, N gaps: positions in the chromosomes as they stand now, one pass per chromosome
    n_scaffolds = n_scaffolds if n_scaffolds is not None else [600, 1500, 4000][j % 3]
    n_tail = n_tail if n_tail is not None else n_scaffolds
    n_gaps = n_gaps if n_gaps is not None else 2 * n_scaffolds
    sizes = np.array([t.size() for t in tabs], dtype=np.int64)
    csum = np.concatenate(([0], np.cumsum(sizes)))

    def draw(n):
        g = np.sort(rng.integers(0, int(csum[-1]), size=int(n)))
        c = np.searchsorted(csum, g, side="right") - 1
        return c, g - csum[c]
    events_by_chrom = [[] for _ in range(n_chrom)]
    for c, x in zip(*draw(max(0, n_scaffolds - n_chrom))):
        events_by_chrom[int(c)].append((int(x), 0, "cut"))
    lo_t, hi_t = tail_bp
    for (c, x), ln in zip(zip(*draw(n_tail)), rng.integers(lo_t, hi_t + 1, size=n_tail)):
        events_by_chrom[int(c)].append((int(x), int(ln), "tail"))
    gap_len = np.where(rng.random(n_gaps) < 0.6, 100, rng.integers(10, 5001, size=n_gaps))
    gap_del = rng.integers(0, 2001, size=n_gaps)
    for (c, x), gl, gd in zip(zip(*draw(n_gaps)), gap_len, gap_del):
        events_by_chrom[int(c)].append((int(x), int(gd), ("gap", int(gl))))
    records, tails = [], []
    for c, t in enumerate(tabs):
        rows = t.p
        starts = [0]
        for r in rows:
            starts.append(starts[-1] + r[1])
        size = starts[-1]
        cur, at = [], 0
        for x, ln, kind in sorted(events_by_chrom[c], key=lambda e: e[0]):
            if x < at + 50 or x + ln + 50 > size:                        # overlaps the event before, or the chromosome's end: dropped
                continue
            cur += _slice_rows(rows, starts, at, x)
            if kind == "cut":
                records.append(cur)
                cur = []
            elif kind == "tail":
                tails.append(_slice_rows(rows, starts, x, x + ln))
            else:
                cur.append([0, kind[1], NRUN, 0])
            at = x + ln
        cur += _slice_rows(rows, starts, at, size)
        records.append(cur)
    records = [r for r in records if sum(x[1] for x in r) > 0] + tails
    rec_len = np.array([sum(x[1] for x in r) for r in records], dtype=np.uint64)
    names = [f"scaffold_{i + 1}" for i in range(len(records))]
    return rec_len, _pieces_array([x for r in records for x in r]), names


def _mix64(x):
    "the device generator's mixer (csrc/ntsynt_hip.hip mix64) on uint64 arrays"
    x = x.copy()
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xff51afd7ed558ccd)
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xc4ceb9fe1a85ec53)
        x ^= x >> np.uint64(33)
    return x


_PHI, _C2, _CELL = np.uint64(0x9E3779B97F4A7C15), np.uint64(0xD1B54A32D192ED03), np.uint64(0xA24BAED4963EE407)
_SALT_LINE, _SALT_SINE, _SALT_SAT = 0x4C494E454C494E45, 0x53494E4553494E45, 0x5341544553415445


def _layer(base, s, seed, salt, cell_log2, len_full, len_min, prob, nfam, div_min, div_max):
    "csrc/ntsynt_hip.hip synth_layer on arrays: overwrite `base` where ancestor coordinate s lies in a repeat copy of this layer"
    if prob == 0:
        return
    u = np.uint64
    cell = s >> u(cell_log2)
    o = (s & u((1 << cell_log2) - 1)).astype(np.int64)
    h = _mix64(u(seed ^ salt) + cell * _CELL)
    ln = len_min + (((h >> u(8)) & u(0xFFFF)) % u(len_full - len_min + 1)).astype(np.int64)
    off = (((h >> u(24)) & u(0xFFFFF)).astype(np.int64)) % ((1 << cell_log2) - ln + 1)
    hit = ((h & u(255)) < u(prob)) & (o >= off) & (o < off + ln)
    if not hit.any():
        return
    idx = (o - off)[hit]
    hh, ss, ll = h[hit], s[hit], ln[hit]
    fam = ((hh >> u(44)) & u(0xFF)) % u(nfam)
    rev = ((hh >> u(52)) & u(1)).astype(bool)
    lvl = ((hh >> u(53)) & u(15)).astype(np.int64)
    div = div_min + lvl * (div_max - div_min) // 15
    ci = np.where(rev, len_full - 1 - idx, len_full - ll + idx).astype(np.uint64)
    cb = _mix64(u(seed ^ salt ^ 0x5555555555555555) + ((fam << u(32)) + ci) * _PHI) & u(3)
    cb = np.where(rev, u(3) - cb, cb)
    y = _mix64(u(seed ^ salt ^ 0xAAAAAAAAAAAAAAAA) + ss * _C2)
    sub = (cb + u(1) + (y >> u(32)) % u(3)) & u(3)
    base[hit] = np.where((y & u(1023)).astype(np.int64) < div, sub, cb)


def _anc_bases(s, seed_anc, rep):
    "ancestor base at coordinates s (uint64 array): synth_ancestor_base"
    with np.errstate(over="ignore"):
        base = _mix64(np.uint64(seed_anc) + s * _PHI) & np.uint64(3)
        if rep:
            _layer(base, s, seed_anc, _SALT_LINE, rep["line_cell_log2"], rep["line_len"], rep["line_min_len"], rep["line_prob_256"], rep["line_families"],
                   rep["div_min_1024"], rep["div_max_1024"])
            _layer(base, s, seed_anc, _SALT_SINE, rep["sine_cell_log2"], rep["sine_len"], rep["sine_len"], rep["sine_prob_256"], rep["sine_families"],
                   rep["div_min_1024"], rep["div_max_1024"])
    return base


def _tandem_bases(s, fam, seed_anc, rep):
    u = np.uint64
    with np.errstate(over="ignore"):
        b = _mix64(u(seed_anc ^ _SALT_SAT) + (u(fam << 32) + s % u(rep["sat_unit"])) * _PHI) & u(3)
        y = _mix64(u(seed_anc ^ _SALT_SAT ^ 0xAAAAAAAAAAAAAAAA) + (u(fam << 40) ^ s) * _C2)
        sub = (b + u(1) + (y >> u(32)) % u(3)) & u(3)
    return np.where((y & u(1023)) < u(rep["sat_div_1024"]), sub, b)


def plan_bases(plan, seed_ancestor, seed_genome, substitution_rate, rep=None):
    """What nts_genome_synth_plan[_ex] generates, evaluated with numpy (tests only: small plans): codes 0..3 = A, C, G, T, 4 = N."""
    rec_len, pieces = plan[0], plan[1]
    n = int(rec_len.sum())
    out = np.empty(n, dtype=np.uint8)
    thr = np.uint64(int(substitution_rate * 4294967296.0))
    with np.errstate(over="ignore"):
        for pc in pieces:
            d, ln, src, fl = int(pc["dst"]), int(pc["len"]), int(pc["src"]), int(pc["flags"])
            if fl & NRUN:
                out[d:d + ln] = 4
                continue
            off = np.arange(ln, dtype=np.uint64)
            sidx = (np.uint64(src + ln - 1) - off) if fl & REVCOMP else (np.uint64(src) + off)
            if fl & NOVEL:
                base = _mix64(np.uint64(seed_genome ^ 0x5bd1e995a7c3f1d7) + sidx * np.uint64(0x9E3779B97F4A7C15)) & np.uint64(3)
            elif fl & TANDEM:
                base = _tandem_bases(sidx, int(pc["reserved"]), seed_ancestor, rep)
            else:
                base = _anc_bases(sidx, seed_ancestor, rep)
            if fl & REVCOMP:
                base = np.uint64(3) - base
            i = np.uint64(d) + off
            y = _mix64(np.uint64(seed_genome) ^ (i * np.uint64(0xD1B54A32D192ED03)))
            hit = (y & np.uint64(0xFFFFFFFF)) < thr
            sub = (base + np.uint64(1) + (y >> np.uint64(32)) % np.uint64(3)) & np.uint64(3)
            out[d:d + ln] = np.where(hit, sub, base).astype(np.uint8)
    return out
