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


def test_odd_bytes_across_many_tiles(ctx, tmp_path):
    """the register path of the parse kernels (whole 16 KiB tiles of sequence lines): lines of every width, LF and CRLF mixed, blanks
    and tabs at the ends of lines and inside them, form feeds, control bytes, '>' in the middle of a line, lower case, N runs,
    empty lines, a header every now and then -- device parse == host reader, bases and faidx columns"""
    rng = np.random.default_rng(20260929)
    out = []
    for r in range(40):
        out.append(b">rec%d some text\t here\n" % r if r % 3 else b">rec%d\r\n" % r)
        for _ in range(int(rng.integers(1, 400))):
            width = int(rng.integers(1, 200))
            line = bytearray(rng.choice(np.frombuffer(b"ACGTacgtNnRYU", np.uint8), size=width).tobytes())
            what = int(rng.integers(0, 12))
            if what == 0:
                line += b"  \t"
            elif what == 1:
                line[:0] = b" \t "
            elif what == 2 and width > 4:
                line[width // 2] = ord(" ")
            elif what == 3 and width > 4:
                line[width // 3] = ord(">")
            elif what == 4 and width > 4:
                line[width // 2] = 1
            elif what == 5:
                line += b"\x0c"
            elif what == 6:
                line = bytearray(b"")
            elif what == 7 and width > 6:
                line[2:5] = b"\t\t\t"
            out.append(bytes(line) + (b"\r\n" if rng.random() < 0.3 else b"\n"))
    raw = b"".join(out)
    assert len(raw) > 40 * 16384
    p = tmp_path / "odd.fa"
    p.write_bytes(raw)
    _same(ctx, str(p))
    q = tmp_path / "odd_no_final_newline.fa"
    q.write_bytes(raw.rstrip(b"\r\n") + b"ACGT")
    _same(ctx, str(q))


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
