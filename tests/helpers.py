"""Shared helpers for the parity tests: random sequences with the edge cases the domain has
(N runs, lower case, ragged / empty / short records) in both containers (oracle and HIP)."""
import numpy as np

from oracle import nts_oracle as O


def random_records(rng, lengths, n_frac=0.01, lower_frac=0.05, n_runs=True):
    seqs = []
    for ln in lengths:
        a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=ln)].copy()
        if ln and n_frac > 0:
            hit = rng.random(ln) < n_frac * 0.2
            a[hit] = ord("N")
            if n_runs and ln > 200:
                for _ in range(max(1, int(ln * n_frac / 200))):
                    st = int(rng.integers(0, ln - 50))
                    a[st:st + int(rng.integers(1, 400))] = ord("N")
        if ln and lower_frac > 0:
            for _ in range(max(1, ln // 5000)):
                st = int(rng.integers(0, max(1, ln - 10)))
                seg = a[st:st + int(rng.integers(1, 300))]
                seg[seg != ord("N")] |= 0x20
        seqs.append(a.tobytes())
    return seqs


def to_oracle(names, seqs):
    return O.Genome(names, seqs)


def to_device(ctx, names, seqs):
    from ntsynt_amd.device import Genome
    lens = np.array([len(s) for s in seqs], dtype=np.uint64)
    off = np.zeros(len(seqs), dtype=np.uint64)
    if len(seqs):
        off[1:] = np.cumsum(lens[:-1])
    blob = np.frombuffer(b"".join(seqs), dtype=np.uint8)
    return Genome(ctx, names, blob, off, lens)


def oracle_flat(mins):
    "oracle per-record minimizers -> flat (h1, rec, pos) arrays like the device list"
    h = np.concatenate([m[0] for m in mins]) if mins else np.zeros(0, np.uint64)
    p = np.concatenate([m[1] for m in mins]) if mins else np.zeros(0, np.uint64)
    r = np.concatenate([np.full(len(m[0]), i, dtype=np.uint32) for i, m in enumerate(mins)]) \
        if mins else np.zeros(0, np.uint32)
    return h, r, p
