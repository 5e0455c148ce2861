"""FASTA ingest for the HIP path (plain or gzip, single- or multi-line, LF or CRLF) and the `.fai` and
minimizer-TSV writers.

Stands in for btllib::SeqReader(LONG_MODE) as used at src/ntsynt_make_common_bf.cpp:32-36,125-131, for
`samtools faidx` (bin/ntsynt_run_pipeline.smk:48-53) and for indexlr's text output (smk:81-85).  Record id =
header up to the first whitespace.  Bases are kept as written; case folding happens on the GPU (k_encode).
The parsing and formatting run in the native library (nts_fasta_read / nts_write_indexlr_tsv in
csrc/nts_hostio.cpp); `read_fasta_numpy` is an independent numpy statement of the same rules, kept for tests."""
import ctypes
import gzip
import os

import numpy as np

from . import _lib


class FastaRecords:
    def __init__(self, names, seq, rec_off, rec_len, fai_rows=None, native=None):
        self.names = names
        self.seq = seq            # uint8, records concatenated, no separators
        self.rec_off = rec_off    # uint64
        self.rec_len = rec_len    # uint64
        self.fai_rows = fai_rows  # [(name, length, offset, linebases, linewidth)]
        self._native = native     # _lib.Fasta owning the buffers behind seq/rec_off/rec_len

    @property
    def total_bp(self):
        return int(self.rec_len.sum())

    def record_bytes(self, i):
        o, n = int(self.rec_off[i]), int(self.rec_len[i])
        return self.seq[o:o + n]

    def __del__(self):
        if getattr(self, "_native", None) is not None:
            try:
                _lib.load().nts_fasta_free(ctypes.byref(self._native))
            except Exception:
                pass
            self._native = None


def read_fasta(path):
    """Native reader (csrc/nts_hostio.cpp).  The arrays are views of the library's buffers, released with the
    FastaRecords object."""
    lib = _lib.load()
    f = _lib.Fasta()
    rc = lib.nts_fasta_read(os.fsencode(path), ctypes.byref(f))
    if rc == -74:                                             # NTS_EFORMAT
        raise ValueError(f"{path!r} is not a FASTA file (a FASTA file starts with a '>' header; FASTQ is not accepted)")
    if rc != 0:
        raise OSError(f"cannot read FASTA file {path!r} (code {rc})")
    n_rec, n = int(f.n_rec), int(f.n)
    seq = np.ctypeslib.as_array(f.seq, shape=(max(n, 1),))[:n]
    rec_off = np.ctypeslib.as_array(f.rec_off, shape=(max(n_rec, 1),))[:n_rec]
    rec_len = np.ctypeslib.as_array(f.rec_len, shape=(max(n_rec, 1),))[:n_rec]
    raw = ctypes.string_at(f.names, int(f.names_bytes))
    names = [x.decode() for x in raw.split(b"\0")[:n_rec]]
    fo = np.ctypeslib.as_array(f.fai_offset, shape=(max(n_rec, 1),))[:n_rec]
    fb = np.ctypeslib.as_array(f.fai_linebases, shape=(max(n_rec, 1),))[:n_rec]
    fw = np.ctypeslib.as_array(f.fai_linewidth, shape=(max(n_rec, 1),))[:n_rec]
    fai = [(names[i], int(rec_len[i]), int(fo[i]), int(fb[i]), int(fw[i])) for i in range(n_rec)]
    return FastaRecords(names, seq, rec_off, rec_len, fai, native=f)


def _records_from_native(f, with_seq):
    n_rec, n = int(f.n_rec), int(f.n)
    seq = np.ctypeslib.as_array(f.seq, shape=(max(n, 1),))[:n] if with_seq else None
    rec_off = np.ctypeslib.as_array(f.rec_off, shape=(max(n_rec, 1),))[:n_rec]
    rec_len = np.ctypeslib.as_array(f.rec_len, shape=(max(n_rec, 1),))[:n_rec]
    raw = ctypes.string_at(f.names, int(f.names_bytes))
    names = [x.decode() for x in raw.split(b"\0")[:n_rec]]
    fo = np.ctypeslib.as_array(f.fai_offset, shape=(max(n_rec, 1),))[:n_rec]
    fb = np.ctypeslib.as_array(f.fai_linebases, shape=(max(n_rec, 1),))[:n_rec]
    fw = np.ctypeslib.as_array(f.fai_linewidth, shape=(max(n_rec, 1),))[:n_rec]
    fai = [(names[i], int(rec_len[i]), int(fo[i]), int(fb[i]), int(fw[i])) for i in range(n_rec)]
    return FastaRecords(names, seq, rec_off, rec_len, fai, native=f)


def read_fasta_device(ctx, path):
    """FASTA file -> (resident Genome, FastaRecords without bases) with the parse on the GPU (nts_genome_from_fasta,
    csrc/nts_fasta_dev.inc): the host reads only the header lines."""
    from .device import Genome
    f = _lib.Fasta()
    h = _lib.c_vp()
    rc = ctx.lib.nts_genome_from_fasta(ctx.h, os.fsencode(path), ctypes.byref(h), ctypes.byref(f))
    if rc == -74:
        raise ValueError(f"{path!r} is not a FASTA file: {ctx.lib.nts_last_error(ctx.h).decode()}")
    ctx.check(rc, "nts_genome_from_fasta")
    recs = _records_from_native(f, with_seq=False)
    g = Genome.__new__(Genome)
    g.ctx, g.h, g.names = ctx, h, recs.names
    g.rec_off = np.array(recs.rec_off, dtype=np.uint64)
    g.rec_len = np.array(recs.rec_len, dtype=np.uint64)
    g.n_bytes = int(f.n)
    g.recs = recs
    return g, recs


def write_indexlr_tsv_kmers(path, recs, h1, rec, pos, k, kmers):
    "write_indexlr_tsv for records whose bases are not on the host: `kmers` (uint8, len(h1) * k) from Minimizers.kmers(), or None"
    h1 = np.ascontiguousarray(h1, dtype=np.uint64)
    rec = np.ascontiguousarray(rec, dtype=np.uint32)
    pos = np.ascontiguousarray(pos, dtype=np.uint64)
    kp = None
    if kmers is not None:
        kmers = np.ascontiguousarray(kmers, dtype=np.uint8)
        if kmers.size != h1.size * int(k):
            raise ValueError("kmers must hold k bytes per minimizer")
        kp = kmers.ctypes.data
    rc = _lib.load().nts_write_indexlr_tsv_kmers(os.fsencode(path), ctypes.byref(recs._native), h1.ctypes.data, rec.ctypes.data,
                                                 pos.ctypes.data, h1.size, int(k), kp)
    if rc != 0:
        raise OSError(f"cannot write {path!r} (code {rc})")


def write_indexlr_tsv(path, recs, h1, rec, pos, k, with_seq=True):
    """`indexlr --long --pos [--seq]` text: one line per FASTA record."""
    if recs._native is None:
        raise ValueError("write_indexlr_tsv needs records read by read_fasta()")
    h1 = np.ascontiguousarray(h1, dtype=np.uint64)
    rec = np.ascontiguousarray(rec, dtype=np.uint32)
    pos = np.ascontiguousarray(pos, dtype=np.uint64)
    rc = _lib.load().nts_write_indexlr_tsv(os.fsencode(path), ctypes.byref(recs._native), h1.ctypes.data, rec.ctypes.data,
                                           pos.ctypes.data, h1.size, int(k), 1 if with_seq else 0)
    if rc != 0:
        raise OSError(f"cannot write {path!r} (code {rc})")


def _load_bytes(path):
    if path.endswith(".gz"):
        with gzip.open(path, "rb") as fh:
            return np.frombuffer(fh.read(), dtype=np.uint8)
    return np.fromfile(path, dtype=np.uint8)


def read_fasta_numpy(path):
    "numpy statement of the same parsing rules (tests compare it with the native reader)"
    data = _load_bytes(path)
    n = data.size
    if n == 0:
        return FastaRecords([], np.zeros(0, np.uint8), np.zeros(0, np.uint64), np.zeros(0, np.uint64), [])
    gt = np.flatnonzero(data == ord(">"))
    if gt.size:
        at_line_start = np.ones(gt.size, dtype=bool)
        nz = gt > 0
        at_line_start[nz] = data[gt[nz] - 1] == 10
        gt = gt[at_line_start]
    nl = np.flatnonzero(data == 10)
    idx = np.searchsorted(nl, gt)
    hdr_end = np.full(gt.size, n, dtype=np.int64)
    has_nl = idx < nl.size
    hdr_end[has_nl] = nl[idx[has_nl]]
    names = []
    for s, e in zip(gt.tolist(), hdr_end.tolist()):
        fields = bytes(data[s + 1:e]).split()
        names.append(fields[0].decode() if fields else "")
    keep = (data != 10) & (data != 13)
    # white space that runs up to the start or the end of its line goes (csrc/nts_fasta_dev.inc fa_is_base)
    blank = np.isin(data, np.array([32, 9, 11, 12, 13], dtype=np.uint8))
    pos = np.arange(n, dtype=np.int64)
    nxt = np.minimum.accumulate(np.where(~blank, pos, n)[::-1])[::-1]            # next byte at or after i that is not blank
    prv = np.maximum.accumulate(np.where(~blank, pos, -1))                        # last byte at or before i that is not blank
    trailing = blank & ((nxt == n) | (data[np.minimum(nxt, n - 1)] == 10))
    leading = blank & ((prv < 0) | (data[np.maximum(prv, 0)] == 10))
    keep &= ~(trailing | leading)
    if gt.size == 0:
        keep[:] = False
    else:
        keep[:gt[0]] = False
        for s, e in zip(gt.tolist(), hdr_end.tolist()):
            keep[s:e + 1] = False
    csum = np.concatenate(([0], np.cumsum(keep, dtype=np.uint64)))
    seq_start = np.minimum(hdr_end + 1, n)
    seq_stop = np.concatenate((gt[1:], [n])) if gt.size else np.zeros(0, dtype=np.int64)
    rec_off = csum[seq_start].astype(np.uint64)
    rec_len = (csum[seq_stop] - csum[seq_start]).astype(np.uint64)
    seq = data[keep]
    fai = []
    for name, s0, s1, ln in zip(names, seq_start.tolist(), seq_stop.tolist(), rec_len.tolist()):
        j = np.searchsorted(nl, s0)
        line_end = int(nl[j]) if j < nl.size and nl[j] < s1 else s1
        width = (line_end - s0 + 1) if (j < nl.size and nl[j] < s1) else (s1 - s0)
        bases = int(csum[line_end] - csum[s0])
        if ln == 0:
            bases = width = 0
        fai.append((name, int(ln), int(s0), int(bases), int(width)))
    return FastaRecords(names, seq, rec_off, rec_len, fai)


def write_fai(path, recs):
    "`samtools faidx` index: NAME LENGTH OFFSET LINEBASES LINEWIDTH"
    with open(path, "w", encoding="utf-8") as fh:
        for row in recs.fai_rows:
            fh.write("\t".join(str(x) for x in row) + "\n")


def basename(path):
    "genome identity everywhere = basename of the FASTA (smk:42; bin/ntsynt_synteny.py:137)"
    return os.path.basename(path)


def read_indexlr_tsv(path):
    """ntJoin's read_minimizers on an `indexlr --long --pos [--seq]` file (stage-3 input of the reference, bin/ntsynt_run.py:12):
    (record ids of all lines, h1, pos, line number of every token), arrays in file order.  Native parser (nts_read_indexlr_tsv)."""
    lib = _lib.load()
    t = _lib.MxTsv()
    rc = lib.nts_read_indexlr_tsv(os.fsencode(path), ctypes.byref(t))
    if rc == -74:
        raise ValueError(f"{path!r}: a minimizer token without a position (indexlr must run with --pos)")
    if rc != 0:
        raise OSError(f"cannot read minimizer TSV {path!r} (code {rc})")
    try:
        n, nl = int(t.n), int(t.n_lines)
        names = [x.decode() for x in ctypes.string_at(t.names, int(t.names_bytes)).split(b"\0")[:nl]]
        h1 = np.ctypeslib.as_array(t.h1, shape=(max(n, 1),))[:n].copy()
        pos = np.ctypeslib.as_array(t.pos, shape=(max(n, 1),))[:n].copy()
        line = np.ctypeslib.as_array(t.line, shape=(max(n, 1),))[:n].copy()
    finally:
        lib.nts_mx_tsv_free(ctypes.byref(t))
    return names, h1, pos, line
