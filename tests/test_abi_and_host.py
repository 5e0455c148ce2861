"""CPU-side checks: the C-ABI library loads and exports every symbol the header declares; FASTA ingest,
.fai writer, CLI validation/defaults (bin/ntSynt:86-120) and the product's failure mode without a GPU."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from ntsynt_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "ntsynt_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(nts_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 35
    lib = _lib.load()
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert declared == bound, (declared - bound, bound - declared)
    for name in declared:
        assert getattr(lib, name) is not None


def test_bf_size_host_call_matches_reference_arithmetic():
    from ntsynt_amd.device import bf_size_bytes
    assert bf_size_bytes(29058289, 0.025) == (143467638, 143467640)
    assert bf_size_bytes(100000000, 0.025)[0] == 493723627
    assert bf_size_bytes(3000000000, 0.025)[0] == 14811708827


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from ntsynt_amd.device import Context, NtsError
    with pytest.raises(NtsError, match="no CPU fallback"):
        Context(0)


def test_fasta_reader_and_fai(tmp_path):
    from ntsynt_amd import fasta as fa
    p = tmp_path / "x.fa"
    p.write_bytes(b">chr1 some description\nACGTACGTAC\nGTACGTNNNN\nAC\n>chr2\n\n>chr3\tx\nacgtnACGT\n")
    r = fa.read_fasta(str(p))
    assert r.names == ["chr1", "chr2", "chr3"]
    assert r.rec_len.tolist() == [22, 0, 9]
    assert bytes(r.record_bytes(0)) == b"ACGTACGTACGTACGTNNNNAC"
    assert bytes(r.record_bytes(2)) == b"acgtnACGT"
    fa.write_fai(str(tmp_path / "x.fa.fai"), r)
    rows = [l.split("\t") for l in open(tmp_path / "x.fa.fai").read().splitlines()]
    assert rows[0] == ["chr1", "22", "23", "10", "11"]                   # samtools faidx columns
    assert rows[2][:2] == ["chr3", "9"] and rows[2][3:] == ["9", "10"]
    # gzip + single-line + CRLF
    import gzip
    q = tmp_path / "y.fa.gz"
    with gzip.open(q, "wb") as fh:
        fh.write(b">a\r\nACGT\r\nAC\r\n>b\r\nTT\r\n")
    r2 = fa.read_fasta(str(q))
    assert r2.names == ["a", "b"] and r2.rec_len.tolist() == [6, 2]
    assert bytes(r2.seq) == b"ACGTACTT"


def test_fasta_reader_matches_oracle_reader(tmp_path):
    from ntsynt_amd import fasta as fa
    from ntsynt_amd import synth
    from oracle import nts_oracle as O
    paths = synth.make_family(str(tmp_path), 2, 300_000, 3, 0.01, seed=4, n_runs=True, soft_mask=True, line_width=70)
    paths += synth.make_family(str(tmp_path), 1, 100_000, 2, 0.0, seed=5, prefix="one")          # single-line records
    for p in paths:
        a, b, c = fa.read_fasta(p), O.read_fasta(p), fa.read_fasta_numpy(p)
        assert a.names == b.names == c.names
        assert a.rec_len.tolist() == b.rec_len.tolist() == c.rec_len.tolist()
        assert a.rec_off.tolist() == c.rec_off.tolist()
        assert bytes(a.seq) == b.blob == bytes(c.seq)
        assert a.fai_rows == c.fai_rows


def test_native_fasta_edge_cases(tmp_path):
    from ntsynt_amd import fasta as fa
    cases = [b"", b"\n \n", b">only\n", b">a\nAC\n\nGT\n>b x y\n", b"junk\n>a\tdesc\r\nAC\r\nG\r\n>b\r\n\r\nT",
             b">a\nACGT",
             # white space at the ends of sequence lines goes, inside a line it stays (an invalid base)
             b">a desc\nACGT  \nAC\t\n \tGGT\r\nTT AC\n\x0c\n>b\n   \nACGTACGT \t \r\nAC\tGT\n>c\nAAAA   "]
    for i, raw in enumerate(cases):
        p = tmp_path / f"c{i}.fa"
        p.write_bytes(raw)
        a, c = fa.read_fasta(str(p)), fa.read_fasta_numpy(str(p))
        assert a.names == c.names, raw
        assert a.rec_len.tolist() == c.rec_len.tolist(), raw
        assert bytes(a.seq) == bytes(c.seq), raw
        assert a.fai_rows == c.fai_rows, raw
        if raw.startswith(b">a desc"):
            from oracle import nts_oracle as O
            assert bytes(a.seq) == b"ACGTACGGTTT ACACGTACGTAC\tGTAAAA" == O.read_fasta(str(p)).blob
    # not FASTA: FASTQ, or text without a single header -> a dedicated error instead of an empty assembly
    for i, raw in enumerate([b"@r1\nACGT\n+\nIIII\n", b"no header at all\nACGT\n"]):
        p = tmp_path / f"bad{i}.fq"
        p.write_bytes(raw)
        with pytest.raises(ValueError, match="not a FASTA file"):
            fa.read_fasta(str(p))


def test_native_tsv_writer_matches_oracle_writer(tmp_path):
    from ntsynt_amd import fasta as fa
    from ntsynt_amd import synth
    from oracle import nts_oracle as O
    from tests.helpers import oracle_flat
    p = synth.make_family(str(tmp_path), 1, 200_000, 3, 0.0, seed=6, soft_mask=True, n_runs=True, line_width=61)[0]
    g = O.read_fasta(p)
    g = O.Genome(g.names + ["empty", "tiny"], [g.record(i) for i in range(3)] + [b"", b"ACGT"])
    synth.write_fasta(str(tmp_path / "x.fa"), [np.frombuffer(g.record(i), dtype=np.uint8) for i in range(5)], names=g.names)
    recs = fa.read_fasta(str(tmp_path / "x.fa"))
    mins = O.minimize(g, 24, 100)
    for with_seq in (True, False):
        O.write_indexlr_tsv(str(tmp_path / "o.tsv"), g, mins, 24, with_seq)
        h1, rec, pos = oracle_flat(mins)
        fa.write_indexlr_tsv(str(tmp_path / "n.tsv"), recs, h1, rec, pos, 24, with_seq)
        assert open(tmp_path / "n.tsv").read() == open(tmp_path / "o.tsv").read()


def _parse(argv):
    from ntsynt_amd import cli
    parser = cli.build_parser()
    args = parser.parse_args(argv)
    return cli.resolve(parser, args), args


def test_cli_divergence_defaults():
    f, a = _parse(["a.fa", "b.fa", "-d", "0.5"])
    assert (a.indel, a.merge, a.w_rounds, a.block_size) == (10000, 10000, [100, 10], 500)
    assert a.prefix == "ntSynt.k24.w1000" and f == ["a.fa", "b.fa"]
    f, a = _parse(["a.fa", "b.fa", "-d", "1"])
    assert (a.indel, a.merge, a.w_rounds, a.block_size) == (50000, 100000, [250, 100], 1000)
    f, a = _parse(["a.fa", "b.fa", "-d", "10.5", "--indel", "7", "--merge", "3w", "-k", "20", "-w", "800"])
    assert (a.indel, a.merge, a.w_rounds, a.block_size) == (7, "3w", [500, 250], 10000)
    assert a.prefix == "ntSynt.k20.w800"


@pytest.mark.parametrize("argv", [
    ["a.fa", "b.fa", "-d", "101"],                 # divergence out of range
    ["a.fa", "b.fa", "-d", "1", "-w", "100"],      # default w_rounds 250 > w
    ["-d", "1"],                                   # no inputs
    ["a.fa", "-d", "1"],                           # fewer than two genomes
    ["a.fa", "b.fa", "--fastas_list", "x", "-d", "1"],
    ["a.fa", "b.fa"],                              # -d is required
])
def test_cli_rejects_like_the_reference(argv):
    with pytest.raises(SystemExit) as e:
        _parse(argv)
    assert e.value.code == 2


def test_cli_missing_file_and_dry_run(tmp_path, capsys):
    from ntsynt_amd import cli
    with pytest.raises(FileNotFoundError):
        cli.main([str(tmp_path / "nope1.fa"), str(tmp_path / "nope2.fa"), "-d", "1"])
    a, b = tmp_path / "a.fa", tmp_path / "b.fa"
    a.write_text(">x\nACGT\n")
    b.write_text(">x\nACGT\n")
    assert cli.main([str(a), str(b), "-d", "1", "-n"]) == 0
    assert "make_common_bf" in capsys.readouterr().out


def test_cli_ends_a_stopped_stage_like_the_reference(tmp_path, monkeypatch, capsys):
    """bin/ntSynt:166-170: a stage that stops -- stage 3's "no paths found" exit (bin/ntsynt_synteny.py:630-632), its duplicate
    --w_rounds check (:597-599), anything raised inside a stage -- ends the run with SubprocessError("ntSynt failed - check the logs
    for the error."), the stage's own message already on the terminal.  Under -n no stage runs, so the duplicate check does not."""
    import subprocess
    import sys
    from ntsynt_amd import cli, pipeline
    a, b = tmp_path / "a.fa", tmp_path / "b.fa"
    a.write_text(">x\nACGT\n")
    b.write_text(">x\nACGT\n")
    argv = [str(a), str(b), "-d", "1"]

    def no_paths(*args, **kw):
        print("Error - no paths found. Try adjusting the specified k/w parameters.")
        sys.exit(1)

    def dies(*args, **kw):
        raise RuntimeError("a stage died")
    monkeypatch.setattr(pipeline, "run", no_paths)
    with pytest.raises(subprocess.SubprocessError, match="ntSynt failed - check the logs for the error.") as e:
        cli.main(argv)
    assert e.value.__cause__ is None and "no paths found" in capsys.readouterr().out
    monkeypatch.setattr(pipeline, "run", dies)
    with pytest.raises(subprocess.SubprocessError) as e:
        cli.main(argv)
    assert isinstance(e.value.__cause__, RuntimeError)
    monkeypatch.setattr(pipeline, "run", lambda *args, **kw: None)
    with pytest.raises(subprocess.SubprocessError):
        cli.main(argv + ["--w_rounds", "100", "100"])
    assert "duplicate values found in w_rounds" in capsys.readouterr().err
    assert cli.main(argv + ["--w_rounds", "100", "100", "-n"]) == 0
    assert cli.main(argv) == 0 and "Done ntSynt!" in capsys.readouterr().out


def test_synthetic_family_is_deterministic(tmp_path):
    from ntsynt_amd import synth
    a = synth.derive_genome(synth.make_ancestor(50_000, 2, seed=3), 0.01, 1, seed=3, micro=3)
    b = synth.derive_genome(synth.make_ancestor(50_000, 2, seed=3), 0.01, 1, seed=3, micro=3)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    anc = synth.make_ancestor(50_000, 2, seed=3)
    g0 = synth.derive_genome(anc, 0.01, 0, seed=3)
    diff = sum(int((x != y).sum()) for x, y in zip(anc, g0)) / 50_000
    assert 0.003 < diff < 0.007           # p/2 substitutions per genome


def test_batch_list_splits_at_record_boundaries():
    """Genome.split_minimizers (host side of nts_genome_concat): the (record, position)-ordered list of a batch is cut
    where the record ids pass a part's first record; parts without minimizers (or without records) give empty lists."""
    from ntsynt_amd.device import Genome
    g = Genome.__new__(Genome)
    g.rec_base = np.array([0, 3, 3, 7, 9], dtype=np.int64)          # parts of 3, 0, 4 and 2 records
    rec = np.array([0, 0, 2, 3, 3, 6, 8, 8, 8], dtype=np.uint32)
    h1 = np.arange(rec.size, dtype=np.uint64) * 11
    pos = np.arange(rec.size, dtype=np.uint64) * 7
    parts = g.split_minimizers(h1, rec, pos)
    assert [p[1].tolist() for p in parts] == [[0, 0, 2], [], [0, 0, 3], [1, 1, 1]]
    assert [p[0].tolist() for p in parts] == [[0, 11, 22], [], [33, 44, 55], [66, 77, 88]]
    assert np.concatenate([p[2] for p in parts]).tolist() == pos.tolist()
    assert all(p[1].dtype == np.uint32 for p in parts)
    empty = g.split_minimizers(h1[:0], rec[:0], pos[:0])
    assert [p[0].size for p in empty] == [0, 0, 0, 0]


# ---- the stage executables of the reference's workflow (ntsynt_amd/stage_cli.py) ---------------------------------------------
def test_stage3_command_line_is_the_reference_s():
    "bin/ntsynt_run.py:10-44: every flag, its default and its type"
    from ntsynt_amd import stage_cli
    a = stage_cli.run_parser().parse_args(["b.fa.k24.w1000.tsv", "a.fa.k24.w1000.tsv", "--fastas", "x/a.fa", "b.fa", "-k", "24", "-w", "1000"])
    assert (a.n, a.p, a.z, a.common, a.repeat, a.btllib_t, a.w_rounds, a.bp, a.collinear_merge, a.simplify_graph, a.m, a.dev, a.interarrivals) == \
        (0, "out", 500, None, None, 4, [100, 10], 500, "1w", False, 90, False, False)
    b = stage_cli.run_parser().parse_args("t1.tsv t2.tsv -k 20 -w 500 --w-rounds 250 100 -p pre --bp 50000 --collinear-merge 100000 -z 1000 "
                                          "--common pre.common.bf --simplify-graph --btllib_t 12 --fastas a.fa b.fa --dev -n 2 -m 80".split())
    assert (b.FILES, b.fastas, b.k, b.w, b.w_rounds, b.p, b.bp, b.collinear_merge, b.z, b.common, b.simplify_graph, b.btllib_t, b.dev, b.n, b.m) == \
        (["t1.tsv", "t2.tsv"], ["a.fa", "b.fa"], 20, 500, [250, 100], "pre", 50000, "100000", 1000, "pre.common.bf", True, 12, True, 2, 80)
    for argv in (["t.tsv", "-k", "24", "-w", "10"], ["t.tsv", "--fastas", "a.fa", "-w", "10"], ["--fastas", "a.fa", "-k", "2", "-w", "10"]):
        with pytest.raises(SystemExit):                                      # --fastas, -k, -w and FILES are required
            stage_cli.run_parser().parse_args(argv)
    assert stage_cli.collinear_merge_bp("3w", 1000) == 3000 and stage_cli.collinear_merge_bp("12345", 1000) == 12345
    with pytest.raises(ValueError):                                          # bin/ntsynt_synteny.py:42
        stage_cli.collinear_merge_bp("3x", 1000)
    assert stage_cli.pair_files(["d/b.fa.k24.w1000.tsv", "a.fa.gz.k20.w100.tsv"], ["x/a.fa.gz", "y/b.fa"]) == ["y/b.fa", "x/a.fa.gz"]
    with pytest.raises(ValueError):
        stage_cli.pair_files(["c.fa.k24.w1000.tsv"], ["a.fa"])


def test_make_common_bf_and_indexlr_command_lines():
    "src/ntsynt_make_common_bf.cpp:46-81; the indexlr options of rule indexlr (smk:81-85) and ntJoin's run_indexlr (S:173-182)"
    from ntsynt_amd import stage_cli
    a = stage_cli.make_common_bf_parser().parse_args(["--genome", "b.fa", "a.fa", "-k", "24"])
    assert (a.genome, a.k, a.fpr, a.p, a.bf, a.t) == (["b.fa", "a.fa"], 24, 0.025, "common_bf", None, 12)
    b = stage_cli.make_common_bf_parser().parse_args("--genome a.fa b.fa c.fa -p pre.common --fpr 0.01 -k 20 -t 48 --bf 1000000".split())
    assert (b.p, b.fpr, b.k, b.t, b.bf) == ("pre.common", 0.01, 20, 48, 1000000)
    with pytest.raises(SystemExit):
        stage_cli.make_common_bf_parser().parse_args(["-k", "24"])            # --genome is required
    c = stage_cli.indexlr_parser().parse_args("-k 24 -w 1000 --long --seq --pos -t 5 -s pre.common.bf -r pre.repeat.bf a.fa".split())
    assert (c.k, c.w, c.long, c.seq, c.pos, c.t, c.s, c.r, c.fasta, c.o) == (24, 1000, True, True, True, 5, "pre.common.bf", "pre.repeat.bf", "a.fa", "/dev/stdout")


def test_minimizer_tsv_reader(tmp_path):
    "ntJoin's read_minimizers on indexlr's text: ids of all lines, tokens hash:pos[:KMER], lines without tokens"
    from ntsynt_amd import fasta as fa
    p = tmp_path / "x.fa.k4.w2.tsv"
    p.write_text("chr1\t18446744073709551615:0:ACGT 456:10:TTTT\nchr2\t\nchr3 desc\t9:5\nchr4\n\n")
    ids, h1, pos, line = fa.read_indexlr_tsv(str(p))
    assert ids == ["chr1", "chr2", "chr3 desc", "chr4", ""]
    assert h1.tolist() == [18446744073709551615, 456, 9] and pos.tolist() == [0, 10, 5] and line.tolist() == [0, 0, 2]
    assert h1.dtype == np.uint64 and line.dtype == np.uint32
    bad = tmp_path / "bad.tsv"
    bad.write_text("chr1\t123 456\n")                                       # indexlr without --pos: ntJoin cannot use it
    with pytest.raises(ValueError):
        fa.read_indexlr_tsv(str(bad))
    with pytest.raises(OSError):
        fa.read_indexlr_tsv(str(tmp_path / "missing.tsv"))
    # a large file goes through several parser threads: same arrays as a line-by-line parse
    rng = np.random.default_rng(1)
    rows = []
    with open(tmp_path / "big.tsv", "w") as fh:
        for r in range(3000):
            n = int(rng.integers(0, 400))
            toks = [(int(rng.integers(0, 2**63)), int(i * 7)) for i in range(n)]
            rows += [(h, ps, r) for h, ps in toks]
            fh.write(f"ctg{r}\t" + " ".join(f"{h}:{ps}:ACGTACGTACGTACGTACGTACGT" for h, ps in toks) + "\n")
    assert os.path.getsize(tmp_path / "big.tsv") > (8 << 20)
    ids, h1, pos, line = fa.read_indexlr_tsv(str(tmp_path / "big.tsv"))
    assert len(ids) == 3000 and h1.tolist() == [r[0] for r in rows] and pos.tolist() == [r[1] for r in rows] and line.tolist() == [r[2] for r in rows]


def test_shard_plan_and_mask_renumbering():
    "fewer genomes than GPUs: groups of ranks per genome, record ranges balanced by bases, masks renumbered for a slice"
    from ntsynt_amd.pipeline import shard_masks, shard_plan
    group_of, ranges = shard_plan([[100, 100, 100, 100], [400], [10, 10, 300, 10, 10, 60]], 8)
    assert group_of == [0, 1, 2, 0, 1, 2, 0, 1]
    by_g = {g: [ranges[r] for r in range(8) if ranges[r][0] == g] for g in range(3)}
    for g, n_rec in ((0, 4), (1, 1), (2, 6)):
        rs = by_g[g]
        assert [x[1] for x in rs] == list(range(len(rs))) and rs[0][2] == 0 and rs[-1][3] == n_rec
        assert all(a[3] == b[2] for a, b in zip(rs, rs[1:])) and all(a[2] <= a[3] for a in rs)
    assert [x[2:] for x in by_g[0]] == [(0, 2), (2, 3), (3, 4)] or sum(b - a for _, _, a, b in by_g[0]) == 4
    assert [x[2:] for x in by_g[1]] == [(0, 0), (0, 0), (0, 1)] or sum(b - a for _, _, a, b in by_g[1]) == 1   # one record: one rank has it
    iv = np.zeros(4, dtype=np.dtype([("rec", np.uint32), ("start", np.uint64), ("end", np.uint64)], align=True))
    iv["rec"] = [0, 2, 3, 5]
    iv["start"] = [5, 6, 7, 8]
    got = shard_masks(iv, 2, 5)
    assert got["rec"].tolist() == [0, 1] and got["start"].tolist() == [6, 7] and iv["rec"].tolist() == [0, 2, 3, 5]
    assert shard_masks([(0, 1, 2), (4, 3, 9)], 3, 6) == [(1, 3, 9)] and shard_masks(None, 0, 1) is None


def test_partition_plan_balances_by_bases_across_genome_boundaries():
    """genomes that do not deal out evenly over the GPUs: the family's records cut into `world` consecutive ranges at record
    boundaries (records are the parallel unit of src/ntsynt_make_common_bf.cpp:128-131,145-153)"""
    from ntsynt_amd import synth
    from ntsynt_amd.pipeline import partition_groups, partition_plan
    # BASELINE configs[2] as bench.py builds it: three 3 Gbp genomes of 24 contigs each, on 8 / 4 / 2 GPUs
    rec_lens = [synth.structural_plan(24, 3_000_000_000 // 24, j, 20240207)[0] for j in range(3)]
    total = sum(int(x.sum()) for x in rec_lens)
    for world in (2, 4, 8):
        parts = partition_plan(rec_lens, world)
        bases = [sum(int(rec_lens[g][a:b].sum()) for g, a, b in ps) for ps in parts]
        assert sum(bases) == total
        assert max(bases) / (total / world) <= 1.05, (world, bases)        # the heaviest rank within 5 % of the mean (3 / 3 / 2 ranks per genome: 1.5)
        seen = [(g, r) for ps in parts for g, a, b in ps for r in range(a, b)]
        assert seen == [(g, r) for g in range(3) for r in range(24)]        # every record once, in family order
        assert all(len({g for g, _, _ in ps}) == len(ps) for ps in parts)  # a rank's parts belong to different genomes
        filters_of, slot_group, n_groups = partition_groups(parts, [24, 24, 24])
        assert sorted({g for row in slot_group for g in row}) == list(range(n_groups))
        assert [len(f) for f in filters_of] == [len(r) for r in slot_group]
    # ragged: a genome of one record, a tiny genome swallowed whole by one rank next to parts of its neighbours, ranks without records
    parts = partition_plan([[100, 100, 100, 100], [400], [10], [10, 10, 300, 10, 10, 60]], 3)
    assert sum(b - a for ps in parts for _, a, b in ps) == 12
    filters_of, slot_group, n_groups = partition_groups(parts, [4, 1, 1, 6])
    for r, fs in enumerate(filters_of):
        whole = [lab for lab, _ in fs if lab[0] == "whole"]
        assert len(whole) <= 1 and all(lab == ("whole", r) for lab in whole)   # a rank's whole genomes share one filter
    covered = sorted(i for _, idx in filters_of[0] for i in idx)
    assert covered == list(range(len(parts[0])))
    parts = partition_plan([[5], [7]], 5)
    assert [len(ps) for ps in parts].count(0) == 3 and sum(len(ps) for ps in parts) == 2
    # one rank: everything, as whole genomes
    assert partition_plan([[1, 2], [3]], 1) == [[(0, 0, 2), (1, 0, 1)]]


def test_bf_file_layout_round_trip_and_rejections(tmp_path):
    """The `{prefix}.common.bf` layout (header table, keys, [HeaderEnd], raw bits: src/ntsynt_make_common_bf.cpp:164 -> btllib's
    save, as recalled -- DESIGN.md 2, u1/u2): what write_bf writes, read_bf reads back bit for bit; files that are not in that
    layout, whose header disagrees with their size, or that hold more than one hash function (ntSynt's filter has one,
    cpp:18-19) are refused with a message instead of being handed to nts_bf_upload."""
    import pytest
    from ntsynt_amd.pipeline import BF_SIGNATURE, bf_header, read_bf, write_bf
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 256, size=1000, dtype=np.uint8)
    p = str(tmp_path / "a.bf")
    write_bf(p, bits, 24)
    got, k = read_bf(p)
    assert k == 24 and np.array_equal(got, bits)
    raw = open(p, "rb").read()
    assert raw.startswith(BF_SIGNATURE.encode() + b"\n") and raw.endswith(bits.tobytes())
    assert raw[:len(raw) - bits.size] == bf_header(bits.size, 24)
    other = str(tmp_path / "b.bf")
    write_bf(other, bits, 32, signature="[BTLKmerBloomFilter_v6]")          # (--bf-signature: the table name is the caller's)
    got, k = read_bf(other)
    assert k == 32 and np.array_equal(got, bits)
    bad = str(tmp_path / "bad.bf")
    open(bad, "wb").write(b">chr1\nACGT\n")
    with pytest.raises(ValueError, match="HeaderEnd"):
        read_bf(bad)
    open(bad, "wb").write(bf_header(999, 24) + bits.tobytes())
    with pytest.raises(ValueError, match="header says 999 bytes"):
        read_bf(bad)
    open(bad, "wb").write(bf_header(bits.size, 24, hash_num=3) + bits.tobytes())
    with pytest.raises(ValueError, match="hash functions"):
        read_bf(bad)


def test_stage3_filter_switch_rules_without_a_gpu(capsys):
    """bin/ntsynt_synteny.py:598-599: `--filter` needs `--repeat` (the reference's ValueError, same text); `--filter Filter` with
    --initial-only is refused before anything touches a device"""
    from ntsynt_amd import stage_cli
    base = ["a.fa.k24.w1000.tsv", "--fastas", "a.fa", "-k", "24", "-w", "1000"]
    a = stage_cli.run_parser().parse_args(base + ["--filter", "Indexlr", "--repeat", "r.bf"])
    assert (a.filter, a.repeat) == ("Indexlr", "r.bf")
    with pytest.raises(ValueError, match="must supply repeat Bloom filter with --repeat"):
        stage_cli.run(base + ["--filter", "Filter"])
    assert stage_cli.run(base + ["--filter", "Filter", "--repeat", "r.bf", "--initial-only"]) == 2
    assert "not with --initial-only" in capsys.readouterr().err
    with pytest.raises(SystemExit):
        stage_cli.run_parser().parse_args(base + ["--filter", "Other"])
