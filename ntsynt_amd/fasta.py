"""FASTA ingest for the HIP path (plain or gzip, single- or multi-line) and the `.fai` writer.

Stands in for btllib::SeqReader(LONG_MODE) as used at src/ntsynt_make_common_bf.cpp:32-36,125-131
and for `samtools faidx` (bin/ntsynt_run_pipeline.smk:48-53).  Record id = header up to the first
whitespace.  Bases are kept as written; case folding happens on the GPU (k_encode)."""
import gzip
import os

import numpy as np


class FastaRecords:
    def __init__(self, names, seq, rec_off, rec_len, fai_rows=None):
        self.names = names
        self.seq = seq            # uint8, records concatenated, no separators
        self.rec_off = rec_off    # uint64
        self.rec_len = rec_len    # uint64
        self.fai_rows = fai_rows  # [(name, length, offset, linebases, linewidth)]

    @property
    def total_bp(self):
        return int(self.rec_len.sum())

    def record_bytes(self, i):
        o, n = int(self.rec_off[i]), int(self.rec_len[i])
        return self.seq[o:o + n]


def _load_bytes(path):
    if path.endswith(".gz"):
        with gzip.open(path, "rb") as fh:
            return np.frombuffer(fh.read(), dtype=np.uint8)
    return np.fromfile(path, dtype=np.uint8)


def read_fasta(path):
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
    # end of each header line
    idx = np.searchsorted(nl, gt)
    hdr_end = np.full(gt.size, n, dtype=np.int64)
    has_nl = idx < nl.size
    hdr_end[has_nl] = nl[idx[has_nl]]
    names = []
    for s, e in zip(gt.tolist(), hdr_end.tolist()):
        fields = bytes(data[s + 1:e]).split()
        names.append(fields[0].decode() if fields else "")
    keep = (data != 10) & (data != 13)
    # drop everything before the first header and the header lines themselves
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
    # faidx columns
    fai = []
    for name, s0, s1, ln in zip(names, seq_start.tolist(), seq_stop.tolist(), rec_len.tolist()):
        j = np.searchsorted(nl, s0)
        line_end = int(nl[j]) if j < nl.size and nl[j] < s1 else s1
        width = (line_end - s0 + 1) if (j < nl.size and nl[j] < s1) else (s1 - s0)
        bases = line_end - s0
        if bases > 0 and data[line_end - 1] == 13:
            bases -= 1
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
