"""The HIP path against the reference's OWN graph stage: tests/golden/refrun/ holds runs of bin/ntsynt_synteny.py's main_synteny
(recorded in the build container by tests/golden/make_golden_refrun.py over stand-ins for ntJoin / igraph / ncls / intervaltree /
bedtools; the assumptions are listed there and in DESIGN.md section 2) on small synthetic families.

  * end to end: `ntsynt_amd.pipeline.run` -- FASTA parse, common Bloom filter, sketches, the graph stage resident in HBM -- writes the
    reference run's pre-collinear-merge and final TSVs byte for byte, the same interarrival lines and the same --dev warnings, with
    both engines (device graph; host-array twin over the device graph build);
  * in lockstep: the device engine (nts_engine_*) next to its host-array twin, state by state, while the twin is held against every
    recorded call of the reference (tests/refrun.py: HostLockstep) -- so each nts_engine_add / _bubbles / _apply / _filter / _erode /
    _blocks result is tied to what run_graph_simplification, find_synteny_blocks, check_for_indels, filter_synteny_blocks,
    filter_minimizers_synteny_blocks, update_list_mx_info, filter_graph_global_flag_overlaps and refine_graph returned there, and the
    GPU's masked re-sketch to the lists the reference's generate_new_minimizers read."""
import contextlib
import io
import os

import pytest

from tests import refrun
from tests.refrun import HostLockstep, Scenario, drive_host

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from ntsynt_amd.device import Context
    c = Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("engine", ["device", "host"])
@pytest.mark.parametrize("name", refrun.scenario_names())
def test_pipeline_writes_the_reference_runs_bytes(ctx, name, engine, in_tmp_cwd):
    from ntsynt_amd import pipeline
    sc = Scenario(name)
    fastas = sc.unpack(str(in_tmp_cwd))
    err = io.StringIO()
    if sc.filter_mode:
        # stage 3 on its own, on the reference's command line (bin/ntsynt_run.py ... --filter <mode> --repeat <bf> --common <bf>): minimizer files,
        # common filter and repeat filter are the run's inputs -- written here by the CPU restatement, which the CPU suite holds against the same
        # recorded run (tests/test_refrun_oracle.py)
        from ntsynt_amd import stage_cli
        from ntsynt_amd.pipeline import write_bf
        from oracle import nts_oracle as O
        m = sc.meta
        k, w = m["k"], m["w"]
        genomes = {p: O.read_fasta(p) for p in fastas}
        common = O.common_bf(genomes, k, 0.025)
        write_bf(f"{sc.prefix}.common.bf", common, k)
        write_bf(f"{sc.prefix}.repeat.bf", sc.repeat_filter([genomes[p] for p in fastas]), k)
        tsvs = []
        for p in fastas:
            tsvs.append(f"{os.path.basename(p)}.k{k}.w{w}.tsv")
            O.write_indexlr_tsv(tsvs[-1], genomes[p], O.minimize(genomes[p], k, w, common), k)
        argv = tsvs + ["-k", str(k), "-w", str(w), "-p", sc.prefix, "--w-rounds"] + [str(x) for x in m["w_rounds"]] + \
            ["--bp", str(m["indel"]), "--collinear-merge", str(m["merge"]), "-z", str(m["z"]), "--common", f"{sc.prefix}.common.bf", "--simplify-graph",
             "--filter", sc.filter_mode, "--repeat", f"{sc.prefix}.repeat.bf", "--dev", "--fastas"] + fastas
        if engine == "host":
            pytest.skip("the stage executable runs the device engine (the host-array engine meets these runs in tests/test_refrun_product.py)")
        with contextlib.redirect_stderr(err), contextlib.redirect_stdout(io.StringIO()):
            assert stage_cli.run(argv) == 0
        for suffix in ("pre-collinear-merge.synteny_blocks.tsv", "synteny_blocks.tsv"):
            with open(f"{sc.prefix}.{suffix}") as fh:
                assert fh.read() == sc.expected(suffix), suffix
        assert [ln for ln in err.getvalue().splitlines() if ln.startswith("WARNING")] == sc.meta["warnings"]
        return
    if sc.stopped:
        # the reference's run ended in an IndexError at S:437 (a round without blocks): so does this one, and like a failed Snakemake rule
        # it leaves no block table behind (the tables up to that point are compared in the lockstep test below)
        with contextlib.redirect_stderr(err), pytest.raises(IndexError, match="ntsynt_synteny.py:437"):
            pipeline.run(fastas, log=lambda *a: None, ctx=ctx, engine=engine, n=sc.min_weight, m=sc.m, dev=True, interarrivals=True, **sc.kwargs())
        assert not os.path.exists(f"{sc.prefix}.synteny_blocks.tsv") and not os.path.exists(f"{sc.prefix}.pre-collinear-merge.synteny_blocks.tsv")
        _same_minimizer_files(sc)
        return
    with contextlib.redirect_stderr(err):
        eng = pipeline.run(fastas, log=lambda *a: None, ctx=ctx, engine=engine, n=sc.min_weight, m=sc.m, dev=True, interarrivals=True, **sc.kwargs())
    assert type(eng).__name__ == ("DeviceSyntenyEngine" if engine == "device" else "SyntenyEngine")
    out = eng.outputs
    assert out[f"{sc.prefix}.pre-collinear-merge.synteny_blocks.tsv"] == sc.expected("pre-collinear-merge.synteny_blocks.tsv")
    assert out[f"{sc.prefix}.synteny_blocks.tsv"] == sc.expected("synteny_blocks.tsv")
    assert sorted(out[f"{sc.prefix}.interarrivals.tsv"].splitlines()) == sorted(sc.expected("interarrivals.tsv").splitlines())
    assert [ln for ln in err.getvalue().splitlines() if ln.startswith("WARNING")] == sc.meta["warnings"]
    _same_minimizer_files(sc)


def _same_minimizer_files(sc):
    "the minimizer TSVs the run wrote are the ones the reference's run read (indexlr restatement in the build container)"
    import hashlib
    for tsv, sha in sc.meta["tsv_sha1"].items():
        with open(tsv, "rb") as fh:
            assert hashlib.sha1(fh.read()).hexdigest() == sha, tsv


@pytest.mark.parametrize("name", refrun.scenario_names())
def test_device_engine_in_lockstep_with_the_reference_run(ctx, name, in_tmp_cwd):
    from ntsynt_amd import fasta as fa
    from ntsynt_amd.device import BloomFilter, Genome, Minimizers, bf_size_bytes, sketch
    from ntsynt_amd.graph import build_graph_device, edge_degrees, walk_paths
    from ntsynt_amd.synteny import SyntenyEngine
    from ntsynt_amd.synteny_device import DeviceSyntenyEngine
    from tests.test_gpu_engine import _blocks_equal, _state_equal
    sc = Scenario(name)
    m = sc.meta
    k, w, rounds = m["k"], m["w"], m["w_rounds"]
    paths = sc.unpack(str(in_tmp_cwd))
    n = len(paths)
    recs = [fa.read_fasta(p) for p in paths]
    genomes = [Genome(ctx, r.names, r.seq, r.rec_off, r.rec_len) for r in recs]
    bf = None
    if m.get("common", True):
        _, nbytes = bf_size_bytes(genomes[sorted(range(n), key=lambda i: paths[i])[0]].total_bp, 0.025)
        bf = BloomFilter(ctx, nbytes, k)
        bf.insert(genomes[0])
        for g in genomes[1:]:
            bf.insert_and(g)
        assert bf.popcount() == m["bf_popcount"] and nbytes == m["bf_bytes"]
    tsvs = [f"{os.path.basename(p)}.k{k}.w{w}.tsv" for p in paths]
    names = [r.names for r in recs]

    # stage 3's repeat filter as ntsynt_amd.pipeline hands it on: filter-out of the refinement sketches (Indexlr), or every list screened (Filter)
    rep = None
    if sc.filter_mode:
        from oracle import nts_oracle as O
        bits = sc.repeat_filter([O.read_fasta(p) for p in paths])
        rep = BloomFilter(ctx, bits.size, k)
        rep.from_numpy(bits)

    def one(i, masks, new_w, refine):
        mx = sketch(ctx, genomes[i], k, new_w, bf, masks, repeat=rep if (refine and sc.filter_mode == "Indexlr") else None)
        if sc.filter_mode == "Filter":
            kept = mx.screened(genomes[i], k, rep)
            mx.free()
            mx = kept
        return mx

    def sketch_np(i, masks, new_w):
        mx = one(i, masks, new_w, masks is not None)
        out = mx.to_numpy()
        mx.free()
        return out

    def sketch_dev(masks_by_asm, new_w):
        return {i: one(i, msk, new_w, True) for i, msk in masks_by_asm.items()}

    os.makedirs("h")
    os.makedirs("d")
    try:
        os.chdir("h")
        host = SyntenyEngine(tsvs, names, k, w, rounds, m["indel"], m["merge"], m["z"], sc.prefix,
                             lambda ls, kp, li: build_graph_device(ctx, ls, kp, li), sketch_np, walk_paths, degree_fn=edge_degrees,
                             n=sc.min_weight, dev=True, interarrivals=True, simplify=sc.simplify, m=sc.m)
        dev = DeviceSyntenyEngine(ctx, tsvs, names, k, w, rounds, m["indel"], m["merge"], m["z"], sc.prefix, sketch_dev, n=sc.min_weight,
                                  simplify=sc.simplify, m=sc.m)
        initial = [sketch_np(i, None, w) for i in range(n)]
        st = {"round": -1, "db": None, "hb": None, "prev_w": w}

        def after(step, h):
            "the device engine takes the step the twin just took; then the two states are compared"
            rnd = st["round"]
            if step == "add":
                if rnd < 0:
                    handles = [Minimizers.from_numpy(ctx, *initial[i]) for i in dev.input_order]
                    dev._add(handles, None)
                elif st["db"]["n"] == 0:                       # a round that starts without a block sketches and adds nothing
                    handles = []
                else:
                    new_w = rounds[rnd]
                    masks = dev._mask_intervals(st["db"], st["prev_w"])
                    handles = dev._sketch_round(masks, new_w)
                    dev._add(handles, dev._spans(st["db"]))
                for mx in handles:
                    mx.free()
                _state_equal(h, dev, f"add, round {rnd}")
            elif step == "simplify":
                dev._simplify_dev(apply_deletions=rnd < 0)
                assert h.stats["bubbles"] == dev.stats["bubbles"]
                _state_equal(h, dev, f"simplify, round {rnd}")
            elif step == "filter":
                last = rnd >= 0 and rounds[rnd] == rounds[-1]
                if (rnd < 0 and h.n > 1) or (rnd >= 0 and (last or h.n > 1)):
                    dev._filter(flag=last)
                _state_equal(h, dev, f"filter, round {rnd}")
            elif step == "erode":
                dev._erode()
                assert h.stats["eroded_edges"] == dev.stats["eroded_edges"]
                _state_equal(h, dev, "erosion")
            elif step == "blocks":
                st["db"] = dev._blocks()
                _state_equal(h, dev, f"blocks, round {rnd}")
                if rnd >= 0:
                    st["prev_w"] = rounds[rnd]
                st["round"] = rnd + 1

        lock = HostLockstep(host, sc, after=after)
        # the device engine's block table is compared inside round_blocks' hook through the twin's block list
        inner = lock.round_blocks

        def round_blocks():
            hb = inner()
            _blocks_equal(host, hb, dev, st["db"], f"block table, round {st['round'] - 1}")
            return hb
        lock.round_blocks = round_blocks
        err = io.StringIO()
        with contextlib.redirect_stderr(err), sc.ends_like_the_reference():
            drive_host(lock, initial)
        out = host.outputs
        assert out[f"{sc.prefix}.pre-collinear-merge.synteny_blocks.tsv"] == sc.expected("pre-collinear-merge.synteny_blocks.tsv")
        assert out[f"{sc.prefix}.synteny_blocks.tsv"] == sc.expected("synteny_blocks.tsv")
        for key in ("bubbles", "unoriented", "indel_cuts", "small_blocks", "eroded_edges"):
            assert host.stats[key] == dev.stats[key], key
        assert lock.checked["paths"] > 0 and lock.checked["valid_minimizers"] > 0
    finally:
        os.chdir("..")
        for g in genomes:
            g.free()
        if bf is not None:
            bf.free()
        if rep is not None:
            rep.free()
