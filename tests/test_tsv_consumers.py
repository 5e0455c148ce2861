"""The synteny TSVs as their downstream readers take them apart (format only): the reference's
analysis_scripts/denovo_synteny_block_stats.py:21-42 (block id, assembly, start, end = columns 0, 1, 3, 4; integer coordinates)
and visualization_scripts/sort_ntsynt_blocks.py:11-12,24-28 (eight columns in the final file -- id, genome, chrom, start, end,
strand, num_mx, reason -- or the first six of the pre-merge file), plus the `.fai` columns both read.  Restated here, not
imported: the product's text (host engine + native writers, no GPU) goes through the same field accesses."""
import os
from collections import namedtuple

from ntsynt_amd import fasta as fa
from ntsynt_amd import synth

SyntenyBlock = namedtuple("SyntenyBlock", ["id", "genome", "chrom", "start", "end", "strand", "num_mx", "reason"])


def test_block_tsvs_parse_like_the_reference_consumers(tmp_path):
    from tests.test_engine_cpu import run_both
    cwd = os.getcwd()
    try:
        paths = synth.make_family(str(tmp_path), 3, 900_000, 3, 0.01, seed=2, micro=6)
        _, got = run_both(tmp_path, paths, 24, 500, [100, 10], 500, 3000, 500)
    finally:
        os.chdir(cwd)
    names = {os.path.basename(p) for p in paths}
    for fname, n_cols in (("p.synteny_blocks.tsv", 8), ("p.pre-collinear-merge.synteny_blocks.tsv", 7)):
        text = got[fname]
        assert text.endswith("\n") and "\r" not in text
        lengths, tallies, last_id = {}, {}, -1
        for line in text.splitlines():
            cols = line.strip().split("\t")
            assert len(cols) == n_cols, (fname, cols)
            # denovo_synteny_block_stats.read_blocks
            block_id, asm, start, end = cols[0], cols[1], int(cols[3]), int(cols[4])
            assert asm in names and 0 <= start < end
            lengths.setdefault(asm, []).append(end - start)
            tallies.setdefault(block_id, set()).add(asm)
            # sort_ntsynt_blocks.sort_blocks
            blk = SyntenyBlock(*cols) if len(cols) == 8 else SyntenyBlock(*cols[:6], None, None)
            assert blk.strand in "+-" and blk.chrom.startswith("chr")
            assert int(blk.id) >= last_id                       # blocks are numbered in file order, rows of a block together
            last_id = int(blk.id)
            if len(cols) >= 7:
                assert int(cols[6]) >= 1                        # number of minimizers
            if len(cols) == 8:
                assert blk.reason in ("None", "id_change", "ori_change", "inconsistent_order", "indel", "merge")
        assert set(lengths) == names
        assert all(len(asms) == len(names) for asms in tallies.values())      # every block lists every assembly once
        assert sorted(int(b) for b in tallies) == list(range(len(tallies)))   # ids 0..n-1


def test_fai_columns_as_the_consumers_read_them(tmp_path):
    "get_genome_size (denovo_synteny_block_stats.py:52-60) and the .fai reader of sort_ntsynt_blocks.py: name, length in columns 0, 1"
    p = synth.make_family(str(tmp_path), 1, 300_000, 3, 0.0, seed=3, line_width=60)[0]
    recs = fa.read_fasta(p)
    out = str(tmp_path / "x.fai")
    fa.write_fai(out, recs)
    total = 0
    for line in open(out, encoding="utf-8"):
        cols = line.strip().split("\t")
        assert len(cols) == 5 and cols[0].startswith("chr")
        total += int(cols[1])
        assert int(cols[3]) == 60 and int(cols[4]) == 61
    assert total == recs.total_bp
