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
