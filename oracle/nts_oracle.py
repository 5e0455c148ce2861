"""ctypes bindings for oracle/nts_oracle.c (CPU restatement of the sketch + Bloom hot path).

TEST INFRASTRUCTURE ONLY -- see the header of nts_oracle.c.  Imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg; never by ntsynt_amd/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_u8p = ctypes.POINTER(ctypes.c_uint8)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_u64 = ctypes.c_uint64


def build(native=False):
    """Compile the oracle with gcc (portable flags; `native=True` adds -march=native)."""
    target = "native" if native else "all"
    subprocess.run(["make", "-s", "-C", _HERE, target], check=True)
    return os.path.join(_HERE, "libnts_oracle_native.so" if native else "libnts_oracle.so")


def _load(native=False):
    path = os.path.join(_HERE, "libnts_oracle_native.so" if native else "libnts_oracle.so")
    src = os.path.join(_HERE, "nts_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        build(native)
    lib = ctypes.CDLL(path)
    lib.nts_o_hash_kmer.argtypes = [ctypes.c_char_p, ctypes.c_uint, _u64p, _u64p]
    lib.nts_o_hash_kmer.restype = ctypes.c_int
    lib.nts_o_hash_all.argtypes = [ctypes.c_char_p, _u64, ctypes.c_uint, _u64p, _u64p]
    lib.nts_o_hash_all.restype = _u64
    lib.nts_o_h1_from_h0.argtypes = [_u64, ctypes.c_uint]
    lib.nts_o_h1_from_h0.restype = _u64
    lib.nts_o_bf_approx_bytes.argtypes = [ctypes.c_longlong, ctypes.c_double]
    lib.nts_o_bf_approx_bytes.restype = ctypes.c_longlong
    lib.nts_o_bf_ctor_bytes.argtypes = [_u64]
    lib.nts_o_bf_ctor_bytes.restype = _u64
    lib.nts_o_bf_ctor_bytes_mode.argtypes = [_u64, ctypes.c_int]
    lib.nts_o_bf_ctor_bytes_mode.restype = _u64
    lib.nts_o_bf_records.argtypes = [_u8p, _u8p, _u64, ctypes.c_char_p, _u64p, _u64p,
                                     ctypes.c_uint32, ctypes.c_uint, ctypes.c_int]
    lib.nts_o_bf_records.restype = None
    lib.nts_o_bf_repeats_seq.argtypes = [_u8p, _u8p, _u64, ctypes.c_char_p, _u64, ctypes.c_uint]
    lib.nts_o_bf_repeats_seq.restype = None
    lib.nts_o_bf_popcount.argtypes = [_u8p, _u64]
    lib.nts_o_bf_popcount.restype = _u64
    lib.nts_o_bf_contains.argtypes = [_u8p, _u64, _u64]
    lib.nts_o_bf_contains.restype = ctypes.c_int
    lib.nts_o_minimize.argtypes = [ctypes.c_char_p, _u64, ctypes.c_uint, ctypes.c_uint, _u8p, _u64,
                                   _u64p, _u64p, _u64]
    lib.nts_o_minimize.restype = _u64
    lib.nts_o_minimize_records.argtypes = [ctypes.c_char_p, _u64p, _u64p, ctypes.c_uint32,
                                           ctypes.c_uint, ctypes.c_uint, _u8p, _u64,
                                           _u64p, _u64p, _u64p, _u64p, ctypes.c_int]
    lib.nts_o_minimize_records.restype = None
    lib.nts_o_minimize2.argtypes = [ctypes.c_char_p, _u64, ctypes.c_uint, ctypes.c_uint, _u8p, _u64, _u8p, _u64,
                                    _u64p, _u64p, _u64]
    lib.nts_o_minimize2.restype = _u64
    lib.nts_o_minimize_records2.argtypes = [ctypes.c_char_p, _u64p, _u64p, ctypes.c_uint32,
                                            ctypes.c_uint, ctypes.c_uint, _u8p, _u64, _u8p, _u64,
                                            _u64p, _u64p, _u64p, _u64p, ctypes.c_int]
    lib.nts_o_minimize_records2.restype = None
    lib.nts_o_minimize_keys.argtypes = [_u64p, _u64, ctypes.c_uint, ctypes.c_int, _u64p, _u64]
    lib.nts_o_minimize_keys.restype = _u64
    return lib


_LIBS = {}


def lib(native=False):
    if native not in _LIBS:
        _LIBS[native] = _load(native)
    return _LIBS[native]


def _p8(a):
    return a.ctypes.data_as(_u8p) if a is not None else None


def _p64(a):
    return a.ctypes.data_as(_u64p)


def hash_kmer(kmer, k=None):
    """(h0, h1) of one k-mer string, or None when it holds a non-ACGT byte."""
    if isinstance(kmer, str):
        kmer = kmer.encode()
    k = k or len(kmer)
    h0, h1 = _u64(), _u64()
    ok = lib().nts_o_hash_kmer(kmer, k, ctypes.byref(h0), ctypes.byref(h1))
    return (h0.value, h1.value) if ok else None


def h1_from_h0(h0, k):
    return lib().nts_o_h1_from_h0(int(h0), k)


def hash_all(seq, k):
    """positions and h0 of every valid k-mer of `seq` (bytes), in order."""
    n = len(seq)
    pos = np.empty(max(n, 1), dtype=np.uint64)
    h0 = np.empty(max(n, 1), dtype=np.uint64)
    cnt = lib().nts_o_hash_all(seq, n, k, _p64(pos), _p64(h0))
    return pos[:cnt].copy(), h0[:cnt].copy()


def bf_approx_bytes(genome_size, fpr):
    "src/ntsynt_make_common_bf.cpp:28-40"
    return lib().nts_o_bf_approx_bytes(int(genome_size), float(fpr))


ROUNDING = {"up": 0, "down": 1, "none": 2}


def bf_ctor_bytes(nbytes, rounding="up"):
    "btllib BloomFilter constructor rounding (SURVEY.md 8(c) u1); `rounding`: the alternatives a btllib reader can settle"
    if rounding == "up":
        return lib().nts_o_bf_ctor_bytes(int(nbytes))
    return lib().nts_o_bf_ctor_bytes_mode(int(nbytes), ROUNDING[rounding])


class Genome:
    """Records of one FASTA held as one bytes blob + offsets (oracle-side container)."""

    def __init__(self, names, seqs):
        self.names = list(names)
        self.rec_len = np.array([len(s) for s in seqs], dtype=np.uint64)
        self.rec_off = np.zeros(len(seqs), dtype=np.uint64)
        if len(seqs):
            self.rec_off[1:] = np.cumsum(self.rec_len[:-1])
        self.blob = b"".join(seqs)

    @property
    def total_bp(self):
        return int(self.rec_len.sum())

    def record(self, i):
        o, n = int(self.rec_off[i]), int(self.rec_len[i])
        return self.blob[o:o + n]


def read_fasta(path):
    """Plain or multi-line FASTA -> Genome; record id = header up to first whitespace (u4).
    Sequence bytes are kept as written (case preserved; hashing is case-insensitive)."""
    names, seqs, cur = [], [], None
    with open(path, "rb") as fh:
        for line in fh:
            if line.startswith(b">"):
                if cur is not None:
                    seqs.append(b"".join(cur))
                hdr = line[1:].split()
                names.append(hdr[0].decode() if hdr else "")
                cur = []
            elif cur is not None:
                cur.append(line.strip())
    if cur is not None:
        seqs.append(b"".join(cur))
    return Genome(names, seqs)


def bf_build(genome, k, bf_bytes, prev=None, threads=1, native=False):
    """One cascade level: prev is None -> level-1 insert (cpp:121-132); else contains->insert
    (cpp:134-160).  bf_bytes is the constructor-rounded byte count.  Returns uint8 array."""
    out = np.zeros(bf_bytes, dtype=np.uint8)
    lib(native).nts_o_bf_records(_p8(prev), _p8(out), bf_bytes, genome.blob, _p64(genome.rec_off),
                                 _p64(genome.rec_len), len(genome.names), k, threads)
    return out


def repeat_bf(genomes, k, bf_bytes):
    """bin/ntsynt_make_repeat_bfs.py:53-69: one repeat filter over all genomes, a fresh per-genome filter each; records in
    order, single thread.  Returns the repeat filter (uint8 array)."""
    rep = np.zeros(bf_bytes, dtype=np.uint8)
    for g in genomes:
        own = np.zeros(bf_bytes, dtype=np.uint8)
        for r in range(len(g.names)):
            rec = g.record(r)
            lib().nts_o_bf_repeats_seq(_p8(own), _p8(rep), bf_bytes, rec, len(rec), k)
    return rep


def bf_popcount(bf):
    return lib().nts_o_bf_popcount(_p8(bf), bf.size)


def bf_fpr(bf):
    "btllib get_fpr() with one hash function: occupancy"
    return bf_popcount(bf) / float(bf.size * 8)


def bf_contains(bf, h0):
    return bool(lib().nts_o_bf_contains(_p8(bf), bf.size, int(h0)))


def common_bf(genomes_by_path, k, fpr, threads=1, bf_bytes=None, native=False, rounding="up"):
    """src/ntsynt_make_common_bf.cpp main(): paths sorted as strings (105-107); size from the
    first (109-118); level 1 then cascade (121-160).  genomes_by_path: {path: Genome}."""
    paths = sorted(genomes_by_path)
    if bf_bytes is None:
        bf_bytes = bf_approx_bytes(genomes_by_path[paths[0]].total_bp, fpr)
    nbytes = bf_ctor_bytes(bf_bytes, rounding)
    bf = bf_build(genomes_by_path[paths[0]], k, nbytes, None, threads, native)
    for p in paths[1:]:
        bf = bf_build(genomes_by_path[p], k, nbytes, bf, threads, native)
    return bf


def minimize(genome, k, w, bf=None, threads=1, native=False, repeat=None):
    """indexlr over every record: list (per record) of (h1 uint64[], pos uint64[]).  bf: filter-in (-s); repeat: filter-out (-r)."""
    n_rec = len(genome.names)
    caps = np.zeros(n_rec + 1, dtype=np.uint64)
    # an upper bound on minimizers per record: every valid k-mer could in principle be one
    per = np.maximum(genome.rec_len.astype(np.int64) - k + 1, 0).astype(np.uint64)
    dense = np.minimum(per, (per // max(w // 8, 1)) + 1024)
    caps[1:] = np.cumsum(dense)
    tot = int(caps[-1])
    out_h = np.empty(max(tot, 1), dtype=np.uint64)
    out_p = np.empty(max(tot, 1), dtype=np.uint64)
    cnt = np.zeros(max(n_rec, 1), dtype=np.uint64)
    lib(native).nts_o_minimize_records2(genome.blob, _p64(genome.rec_off), _p64(genome.rec_len), n_rec,
                                        k, w, _p8(bf), 0 if bf is None else bf.size, _p8(repeat), 0 if repeat is None else repeat.size,
                                        _p64(out_h), _p64(out_p), _p64(caps), _p64(cnt), threads)
    res = []
    for r in range(n_rec):
        c, o = int(cnt[r]), int(caps[r])
        if c > int(caps[r + 1]) - o:  # capacity guess too small: redo this record exactly
            hh = np.empty(c, dtype=np.uint64)
            pp = np.empty(c, dtype=np.uint64)
            rec = genome.record(r)
            lib(native).nts_o_minimize2(rec, len(rec), k, w, _p8(bf), 0 if bf is None else bf.size,
                                        _p8(repeat), 0 if repeat is None else repeat.size, _p64(hh), _p64(pp), c)
            res.append((hh, pp))
        else:
            res.append((out_h[o:o + c].copy(), out_p[o:o + c].copy()))
    return res


def write_indexlr_tsv(path, genome, mins, k, with_seq=True):
    """`indexlr --long --pos [--seq]` text: id \\t h1:pos[:kmer] ...\\n;
    a record with no minimizers prints its id and an empty second column."""
    with open(path, "w", encoding="utf-8") as out:
        for r, name in enumerate(genome.names):
            h, p = mins[r]
            rec = genome.record(r) if with_seq else None
            toks = []
            for hv, pv in zip(h.tolist(), p.tolist()):
                if with_seq:
                    toks.append(f"{hv}:{pv}:{rec[pv:pv + k].decode().upper()}")
                else:
                    toks.append(f"{hv}:{pv}")
            out.write(f"{name}\t{' '.join(toks)}\n")


def minimize_keys(keys, w, strict=False):
    """Positions the window rule of `minimize` emits for a stream of comparison keys (uint64 per position, 2^64-1 = no
    accepted k-mer there).  strict=True swaps the tie rule to `<` (a negative control for the tests, not the restatement)."""
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    out = np.empty(max(keys.size, 1), dtype=np.uint64)
    n = lib().nts_o_minimize_keys(_p64(keys), keys.size, int(w), int(bool(strict)), _p64(out), out.size)
    return out[:n].copy()
