"""FASTA ingest with the parse on the GPU (nts_genome_from_fasta, csrc/nts_fasta_dev.inc) against the host reader
(nts_fasta_read) and its numpy statement: record ids, lengths, offsets, faidx columns, and the resident bases themselves."""
import gzip
import os

import numpy as np
import pytest

from ntsynt_amd import fasta as fa
from ntsynt_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from ntsynt_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def _codes(seq):
    "what the resident genome reads back as: A/C/G/T (U -> T) upper case, anything else N"
    a = np.frombuffer(bytes(seq), dtype=np.uint8).copy() & 0xDF
    out = np.full(a.size, ord("N"), np.uint8)
    for c in b"ACGT":
        out[a == c] = c
    out[a == ord("U")] = ord("T")
    return out


def _same(ctx, path):
    from ntsynt_amd.device import Genome
    g, recs = fa.read_fasta_device(ctx, path)
    host = fa.read_fasta(path)
    assert recs.names == host.names == g.names
    assert recs.rec_len.tolist() == host.rec_len.tolist()
    assert recs.rec_off.tolist() == host.rec_off.tolist()
    assert recs.fai_rows == host.fai_rows
    assert recs.seq is None and g.total_bp == host.total_bp
    n = int(host.seq.size)
    if n:
        assert np.array_equal(g.download(0, n), _codes(host.seq))
        up = Genome(ctx, host.names, host.seq, host.rec_off, host.rec_len)
        for k in (20, 24):
            assert g.valid_kmers(k) == up.valid_kmers(k)
        up.free()
    g.free()
    return host


def test_device_parse_edge_cases(ctx, tmp_path):
    cases = [b"", b"\n \n", b">only\n", b">a\nAC\n\nGT\n>b x y\n", b"junk\n>a\tdesc\r\nAC\r\nG\r\n>b\r\n\r\nT", b">a\nACGT",
             b">a\n>b\n>c\nACGTNNacgtnRYKM\n", b">x y z\nACGU\nacgu", b">a\n" + b"ACGT" * 9000 + b"\n>b\n" + b"T" * 70000]
    for i, raw in enumerate(cases):
        p = tmp_path / f"c{i}.fa"
        p.write_bytes(raw)
        _same(ctx, str(p))


def test_device_parse_families_multiline_crlf_gz(ctx, tmp_path):
    paths = synth.make_family(str(tmp_path), 2, 3_000_000, 3, 0.01, seed=4, n_runs=True, soft_mask=True, line_width=70)
    paths += synth.make_family(str(tmp_path), 1, 2_000_000, 2, 0.0, seed=5, prefix="one")            # single-line records
    paths += synth.make_family(str(tmp_path), 1, 1_500_000, 700, 0.0, seed=6, prefix="frag", line_width=61)   # many short records
    for p in paths:
        _same(ctx, p)
    # CRLF line ends and a gzip copy
    raw = open(paths[0], "rb").read().replace(b"\n", b"\r\n")
    crlf = tmp_path / "crlf.fa"
    crlf.write_bytes(raw)
    _same(ctx, str(crlf))
    gz = tmp_path / "copy.fa.gz"
    with gzip.open(gz, "wb") as fh:
        fh.write(open(paths[0], "rb").read())
    host = _same(ctx, str(gz))
    assert host.total_bp == fa.read_fasta(paths[0]).total_bp


def test_white_space_in_sequence_lines(ctx, tmp_path):
    """blanks, tabs, form feeds at either end of a sequence line are dropped (the reference's reader trims its lines; the oracle's
    strips both ends), inside a line they stay as invalid bases; host reader, device parse and the oracle's reader agree"""
    from oracle import nts_oracle as O
    raw = (b">a desc\nACGT  \nAC\t\n \tGGT\r\nTT AC\n\x0c\n>b\n   \nACGTACGT \t \r\nAC\tGT\n>c\nAAAA   ")
    p = tmp_path / "ws.fa"
    p.write_bytes(raw)
    host = _same(ctx, str(p))
    og = O.read_fasta(str(p))
    assert host.names == og.names == ["a", "b", "c"]
    assert [bytes(host.seq[int(o):int(o) + int(n)]) for o, n in zip(host.rec_off, host.rec_len)] == \
        [b"ACGTACGGTTT AC", b"ACGTACGTAC\tGT", b"AAAA"]
    assert [og.record(i) for i in range(3)] == [b"ACGTACGGTTT AC", b"ACGTACGTAC\tGT", b"AAAA"]
    assert host.fai_rows[0][3:] == (4, 7) and host.fai_rows[2][3:] == (4, 7)         # bases / bytes of the first line


def test_device_parse_rejects_what_is_not_fasta(ctx, tmp_path):
    for i, raw in enumerate([b"@r1\nACGT\n+\nIIII\n", b"no header at all\nACGT\n"]):
        p = tmp_path / f"bad{i}.fq"
        p.write_bytes(raw)
        with pytest.raises(ValueError, match="not a FASTA file"):
            fa.read_fasta_device(ctx, str(p))


def test_kmer_text_from_hbm_matches_host_tsv(ctx, tmp_path):
    "`indexlr --seq` output written from k-mer text gathered in HBM == the writer that reads the host's bases"
    from ntsynt_amd.device import Genome, sketch
    p = synth.make_family(str(tmp_path), 1, 1_000_000, 3, 0.0, seed=8, soft_mask=True, n_runs=True, line_width=80)[0]
    g, recs = fa.read_fasta_device(ctx, p)
    host = fa.read_fasta(p)
    mx = sketch(ctx, g, 24, 100)
    h1, rec, pos = mx.to_numpy()
    km = mx.kmers(g, 24)
    fa.write_indexlr_tsv_kmers(str(tmp_path / "d.tsv"), recs, h1, rec, pos, 24, km)
    fa.write_indexlr_tsv(str(tmp_path / "h.tsv"), host, h1, rec, pos, 24, True)
    assert open(tmp_path / "d.tsv").read() == open(tmp_path / "h.tsv").read()
    fa.write_indexlr_tsv_kmers(str(tmp_path / "d2.tsv"), recs, h1, rec, pos, 24, None)
    fa.write_indexlr_tsv(str(tmp_path / "h2.tsv"), host, h1, rec, pos, 24, False)
    assert open(tmp_path / "d2.tsv").read() == open(tmp_path / "h2.tsv").read()
    mx.free()
    g.free()
