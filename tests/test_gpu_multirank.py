"""More than one rank on a box with ONE GPU: the product's exchange code (csrc/nts_comm.inc: nts_bf_allreduce_and,
nts_mx_allgather -- the reduce-scatter by grouped send/recv, the AND of the received pieces, the in-place all-gather, the
packed list payload) executed with world 2 and 3, plus bench.py's and the pipeline's multi-rank paths end to end.

Real RCCL refuses two ranks on one device, so the ranks (processes sharing GPU 0) reach each other through the test-only
librccl stand-in (tests/rccl_standin/, selected with NTS_RCCL_LIB): same C entry points, same call sequence, device
buffers on both ends.  What this does NOT show is speed or RCCL's own behaviour on xGMI -- that is the driver's
multi-GPU run.  The schedules of the torch.distributed fallback (ntsynt_amd/dist.py) are covered on CPU by
tests/test_dist_gloo.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STANDIN = os.path.join(ROOT, "tests", "rccl_standin", "librccl_standin.so")
WORKER = os.path.join(ROOT, "tests", "multirank_worker.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env(**extra):
    assert os.path.exists(STANDIN), "tests/rccl_standin/librccl_standin.so is not built (__graft_entry__.build())"
    return dict(os.environ, PYTHONPATH=ROOT, NTS_RCCL_LIB=STANDIN, **extra)


def _ranks(world, case, tmp_path, **extra):
    "`world` worker processes on GPU 0; returns their JSON lines"
    id_file = str(tmp_path / f"id_{case}_{world}")
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), id_file, case], env=_env(**extra), cwd=tmp_path,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    for r, p in enumerate(procs):
        try:
            so, se = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, f"rank {r}: {se[-3000:]}"
        outs.append(json.loads([ln for ln in so.splitlines() if ln.startswith("{")][-1]))
    return outs


def _torchrun(n, script_args, env_extra, cwd):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, cwd=cwd, env=_env(MASTER_ADDR="127.0.0.1", **env_extra), capture_output=True, text=True, timeout=900)


@pytest.mark.parametrize("world", [2, 3])
def test_bf_allreduce_and_between_ranks(world, tmp_path):
    """exchange 1 == bitwise AND of the ranks' filters: sizes that do not divide by 16 x world, pieces forced small so the
    reduce-scatter loop iterates (3,000,008 bytes / world in 64 KiB pieces), a rank contributing the identity"""
    outs = _ranks(world, "allreduce", tmp_path, NTS_COMM_PIECE="65536")
    assert all(o["library"] == STANDIN and o["sizes"][-1] == 3_000_008 for o in outs)


@pytest.mark.parametrize("world", [2, 3])
def test_bf_allreduce_and_one_piece(world, tmp_path):
    "same, with the product's piece size (every chunk is one piece)"
    outs = _ranks(world, "allreduce", tmp_path)
    assert len(outs) == world


@pytest.mark.parametrize("world", [3, 4])
def test_bf_allreduce_groups_between_ranks(world, tmp_path):
    """exchange 1 with fewer genomes than ranks (nts_bf_allreduce_groups): AND over genomes of the OR over a genome's shards, one
    W-way reduction kernel per piece on a second stream, pieces forced small so that transfer and reduction alternate between
    the two scratch halves; an all-but-empty result travels as set-bit indices"""
    outs = _ranks(world, "groups", tmp_path, NTS_COMM_PIECE="65536")
    for o in outs:
        assert o["sparse"] == [False, False, False, True, True] and o["rejects_empty_group"]
    os.remove(tmp_path / f"id_groups_{world}")                           # (a second communicator: a second id)
    # the same through the dense all-gather only: a switch of the experiments build of the library (csrc/nts_knobs.h)
    outs = _ranks(world, "groups", tmp_path, NTS_COMM_SPARSE="0", NTS_LIB_VARIANT="experiments")
    assert all(o["sparse"] == [False] * 5 for o in outs)


@pytest.mark.parametrize("world", [2, 3, 4])
def test_bf_allreduce_parts_between_ranks(world, tmp_path):
    """exchange 1 with a family's records shared out by bases across genome boundaries (nts_bf_allreduce_parts): a rank holds one filter
    per genome its range touches (two where the range crosses a boundary, none when it has no record); AND over genomes of the OR over a
    genome's parts; pieces forced small; the lists of exchange 2 likewise several per rank (nts_mx_allgather_ex)"""
    outs = _ranks(world, "parts", tmp_path, NTS_COMM_PIECE="65536")
    for o in outs:
        assert o["sparse"] == [False, False, False, True] and o["rejects_gaps"]


def test_genome_slice_and_list_concat_on_one_rank(tmp_path):
    "nts_genome_slice + nts_mx_concat: the shards of a genome give its filter (OR) and its minimizer list (concatenation) exactly"
    import numpy as np
    sys.path.insert(0, ROOT)
    from ntsynt_amd.device import BloomFilter, Context, Minimizers, bf_size_bytes, sketch
    from ntsynt_amd.pipeline import shard_plan
    from tests.helpers import random_records, to_device
    ctx = Context(0)
    rng = np.random.default_rng(5)
    seqs = random_records(rng, [90_000, 0, 1500, 300_000, 40, 120_000, 2500, 60_000])
    names = [f"r{i}" for i in range(len(seqs))]
    g = to_device(ctx, names, seqs)
    k, w = 24, 200
    _, nbytes = bf_size_bytes(g.total_bp, 0.025)
    whole = BloomFilter(ctx, nbytes, k)
    whole.insert(g)
    full = sketch(ctx, g, k, w, whole).to_numpy()
    for n_shards in (2, 3, 5):
        _, ranges = shard_plan([[len(s) for s in seqs], [1]], 2 * n_shards)      # genome 0 over the even ranks
        cuts = [ranges[r][2:] for r in range(0, 2 * n_shards, 2)]
        assert cuts[0][0] == 0 and cuts[-1][1] == len(seqs) and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
        union = BloomFilter(ctx, nbytes, k)
        parts = []
        for r0, r1 in cuts:
            sub = g.slice(r0, r1)
            assert sub.total_bp == sum(len(s) for s in seqs[r0:r1])
            union.insert(sub)                                                     # OR into the same filter
            parts.append(sketch(ctx, sub, k, w, whole))
            sub.free()
        assert np.array_equal(union.to_numpy(), whole.to_numpy())
        cat = Minimizers.concat(ctx, parts, [c[0] for c in cuts])
        for a, b in zip(cat.to_numpy(), full):
            assert np.array_equal(a, b)
        for m in parts + [cat]:
            m.free()
        union.free()
    ctx.close()


@pytest.mark.parametrize("world", [2, 3])
def test_mx_allgather_between_ranks(world, tmp_path):
    "exchange 2: uneven list counts per rank, an empty list, lists of very different lengths; repeated genome numbers rejected"
    outs = _ranks(world, "allgather", tmp_path)
    assert all(o["rejects_repeats"] for o in outs)


def test_bench_two_ranks(tmp_path):
    r = _torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "c4", "--genomes", "4",
                      "--mbp", "6", "--contigs", "2"], {"NTS_BENCH_BACKEND": "gloo"}, tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0 and out["steps"] == 2
    assert out["config"]["genomes_on_rank0"] == 2 and "c4: 4 synthetic" in out["config"]["workload"]
    assert out["config"]["exchanges"].startswith("libntsynt_hip.so (nts_bf_allreduce_and")
    assert out["config"]["rccl_ranks"] == 2 and out["config"]["rccl_library"] == STANDIN
    assert out["bloom"]["allreduce_and_s"] > 0
    assert "e2e" not in out and "cpu_baseline" not in out          # single-GPU legs only


def _bench_line(cmd, env, cwd):
    r = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks_and_keeps_one_workload_along_the_curve(tmp_path):
    """`python bench.py --gpus 4 --steps 2` -- plain, no torchrun, the driver's command form -- starts four ranks itself and reports
    n_gpus 4 with rccl_ranks 4 on the SAME workload as N = 1 (c3: three genomes, their records shared out by bases across genome
    boundaries), config 4 as the `c4` leg of the same line at both N; a launcher whose WORLD_SIZE disagrees with --gpus is refused"""
    small = ["--steps", "2", "--warmup", "1", "--mbp", "8", "--contigs", "4"]
    env = _env(NTS_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    out = _bench_line([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"] + small, env, tmp_path)
    assert out["n_gpus"] == 4 and out["value"] > 0 and "c3: 3 synthetic" in out["config"]["workload"]
    assert out["config"]["rccl_ranks"] == 4 and out["config"]["rccl_library"] == STANDIN and out["config"]["library"].endswith("libntsynt_hip.so")
    assert "3 genomes over 4 GPUs" in out["config"]["parallelism"] and out["bloom"]["allreduce_and_s"] > 0
    assert out["config"]["balance"]["max_over_mean"] <= 1.35           # (12 records of ~2 Mbp on 4 ranks: 3 each)
    assert out["c4"]["n_gpus"] == 4 and out["c4"]["value_Gbases_s"] > 0 and out["c4"]["allreduce_and_s"] > 0
    # the same family on one rank: the same workload key, the same number of minimizers per step, in both the headline and the c4 leg
    ref = _bench_line([sys.executable, os.path.join(ROOT, "bench.py")] + small +
                      ["--no-e2e", "--no-cpu-baseline", "--no-dense-leg", "--no-cold-leg", "--no-nruns-leg", "--no-c5-leg", "--no-valley-leg"],
                      dict(os.environ, PYTHONPATH=ROOT), tmp_path)
    assert ref["n_gpus"] == 1 and ref["config"]["workload"].split(",")[0] == out["config"]["workload"].split(",")[0]
    assert ref["config"]["bases_per_step"] == out["config"]["bases_per_step"]
    assert out["config"]["minimizers_per_step_all_genomes"] == ref["config"]["minimizers_per_step_rank0"]
    assert ref["c4"]["n_gpus"] == 1 and ref["c4"]["minimizers_per_step_all_genomes"] == out["c4"]["minimizers_per_step_all_genomes"]
    assert abs(ref["c4"]["common_filter_occupancy"] - out["c4"]["common_filter_occupancy"]) < 1e-15
    # a launcher that started another number of ranks than --gpus says: refused, no line
    r = _torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "4"] + small, {"NTS_BENCH_BACKEND": "gloo"}, tmp_path)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "refusing" in r.stderr


def test_bench_with_eight_ranks_the_driver_command_at_small_size(tmp_path):
    """`python bench.py --gpus 8` (the driver's scaling command, here with eight ranks on the visible GPU through the stand-in): three
    genomes shared out over eight ranks by bases (a rank holds up to two parts), config 4's eight genomes one per rank; the lists
    cross exchange 2 packed (12 bytes per minimizer and a record table instead of 20) and add up to the one-rank run's"""
    small = ["--steps", "2", "--warmup", "1", "--mbp", "8", "--contigs", "4"]
    env = _env(NTS_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    out = _bench_line([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"] + small, env, tmp_path)
    assert out["n_gpus"] == 8 and out["value"] > 0 and out["config"]["rccl_ranks"] == 8
    assert "3 genomes over 8 GPUs" in out["config"]["parallelism"]
    x2 = out["config"]["exchange2_bytes_per_step"]
    n_all = out["config"]["minimizers_per_step_all_genomes"]
    assert x2["unpacked_bytes"] == 20 * n_all
    assert 12 * n_all <= x2["packed_bytes"] <= 12 * n_all + 8 * 64 * 12 and x2["packed_over_unpacked"] <= 0.62      # (<= 0.6 x + the record tables)
    assert x2["sent_bytes"] > 0
    assert out["c4"]["n_gpus"] == 8 and out["c4"]["balance"]["max_over_mean"] <= 1.01 and out["c4"]["exchange2_bytes_per_step"]["packed_bytes"] > 0
    ref = _bench_line([sys.executable, os.path.join(ROOT, "bench.py")] + small +
                      ["--no-e2e", "--no-cpu-baseline", "--no-dense-leg", "--no-cold-leg", "--no-nruns-leg", "--no-c5-leg", "--no-valley-leg"],
                      dict(os.environ, PYTHONPATH=ROOT), tmp_path)
    assert n_all == ref["config"]["minimizers_per_step_rank0"]
    assert ref["c4"]["minimizers_per_step_all_genomes"] == out["c4"]["minimizers_per_step_all_genomes"]
    assert ref["config"]["exchange2_bytes_per_step"] is None and ref["config"]["value_speedup_vs_n1"] is None


@pytest.mark.parametrize("world,n_genomes,contigs,extra", [(4, 3, 3, []), (5, 2, 2, []), (3, 2, 1, []), (2, 3, 3, []), (3, 5, 2, []),
                                                           (4, 3, 2, ["--no-common", "--no-simplify-graph"])])
def test_pipeline_with_genomes_that_do_not_deal_out_evenly_matches_single_rank(world, n_genomes, contigs, extra, tmp_path):
    """bin/ntSynt under torchrun with a number of genomes that is no multiple of the ranks (more ranks than genomes, or three genomes
    on two ranks, five on three): the family's records are shared out over the ranks by bases, across genome boundaries
    (pipeline.partition_plan); a rank builds a filter per genome its range touches, OR-ed over a genome's parts and AND-ed across
    genomes in one exchange (nts_bf_allreduce_parts); every round's part lists are gathered and strung together per genome
    (nts_mx_concat) -- byte for byte what one rank writes, filter file and minimizer TSVs included.  (5 ranks on 2 genomes of 2 records:
    a rank with no record at all; 3 ranks on 2 single-record genomes: likewise.)"""
    from ntsynt_amd import synth
    paths = synth.make_family(str(tmp_path), n_genomes, 1_500_000, contigs, 0.01, seed=45, micro=6, n_runs=True)
    one = tmp_path / "one"
    many = tmp_path / "many"
    os.makedirs(one)
    os.makedirs(many)
    args = ["-k", "24", "-w", "500", "-d", "1", "--prefix", "p", "--indel", "5000", "--merge", "20000", "--force"] + extra + paths
    env = dict(os.environ, PYTHONPATH=ROOT)
    subprocess.run([sys.executable, os.path.join(ROOT, "bin", "ntSynt")] + args, cwd=one, env=env, check=True,
                   stdout=subprocess.DEVNULL, timeout=600)
    r = _torchrun(world, [os.path.join(ROOT, "bin", "ntSynt")] + args, {"NTS_DIST_BACKEND": "gloo", "NTS_COMM_PIECE": "262144"}, many)
    assert r.returncode == 0, r.stderr[-3000:]
    names = ["p.synteny_blocks.tsv", "p.pre-collinear-merge.synteny_blocks.tsv"] + ([] if "--no-common" in extra else ["p.common.bf"])
    names += [os.path.basename(p) + ".k24.w500.tsv" for p in paths] + [os.path.basename(p) + ".fai" for p in paths]
    for name in names:
        assert (one / name).read_bytes() == (many / name).read_bytes(), name
    assert len((one / "p.synteny_blocks.tsv").read_text().splitlines()) >= 4
    assert not [n for n in os.listdir(many) if n.startswith(".ntsynt_rank")]


@pytest.mark.parametrize("world", [2, 3])
def test_pipeline_ranks_match_single_rank(world, tmp_path):
    """bin/ntSynt under torchrun with `world` ranks (genomes sharded, nts_bf_allreduce_and, nts_mx_allgather in every round,
    replicated graph stage) == one rank, byte for byte -- the filter file included"""
    from ntsynt_amd import synth
    paths = synth.make_family(str(tmp_path), 4, 1_200_000, 2, 0.01, seed=44, micro=6)
    one = tmp_path / "one"
    many = tmp_path / "many"
    os.makedirs(one)
    os.makedirs(many)
    args = ["-k", "24", "-w", "500", "-d", "1", "--prefix", "p", "--indel", "5000", "--merge", "20000", "--force"] + paths
    env = dict(os.environ, PYTHONPATH=ROOT)
    subprocess.run([sys.executable, os.path.join(ROOT, "bin", "ntSynt")] + args, cwd=one, env=env, check=True,
                   stdout=subprocess.DEVNULL, timeout=600)
    r = _torchrun(world, [os.path.join(ROOT, "bin", "ntSynt")] + args, {"NTS_DIST_BACKEND": "gloo", "NTS_COMM_PIECE": "262144"}, many)
    assert r.returncode == 0, r.stderr[-3000:]
    names = ["p.synteny_blocks.tsv", "p.pre-collinear-merge.synteny_blocks.tsv", "p.common.bf"]
    names += [os.path.basename(p) + ".k24.w500.tsv" for p in paths]
    for name in names:
        assert (one / name).read_bytes() == (many / name).read_bytes(), name
    assert len((one / "p.synteny_blocks.tsv").read_text().splitlines()) >= 8


def test_packed_exchange_on_one_rank_round_trips_every_shape_and_refuses_unordered_lists():
    """nts_mx_allgather_ex with one rank runs the same pack -> payload -> unpack path as with eight: lists with records that hold no
    minimizer (their start index equals their successor's), one list whose positions need 64 bits, an empty list, a list of one -- all
    come back as they went in; a list that is not in record order is refused (the packed form stores one start index per record)"""
    import numpy as np
    from ntsynt_amd.device import Context, Minimizers, allgather_minimizers
    ctx = Context(0)
    try:
        rng = np.random.default_rng(3)
        lists = []
        for case in range(5):
            n = [5000, 1, 0, 777, 40000][case]
            rec = np.sort(rng.choice(np.arange(3, 3 + [40, 1, 1, 900, 7][case]), size=n)).astype(np.uint32)     # gaps in the record ids: records without a minimizer
            pos = rng.integers(0, (1 << 40) if case == 3 else (1 << 31), size=n).astype(np.uint64)
            h1 = rng.integers(0, 1 << 63, size=n).astype(np.uint64)
            lists.append((h1, rec, pos))
        handles = [Minimizers.from_numpy(ctx, *t) for t in lists]
        out = allgather_minimizers(ctx, None, handles, [4, 0, 2, 1, 3], 5, slots=5)
        for gid, t in zip([4, 0, 2, 1, 3], lists):
            got = out[gid].to_numpy()
            for a, b in zip(got, t):
                assert np.array_equal(a, b), gid
        for m in out + handles:
            m.free()
        bad = Minimizers.from_numpy(ctx, np.arange(10, dtype=np.uint64), np.array([0, 0, 1, 1, 2, 1, 2, 2, 3, 3], np.uint32), np.arange(10, dtype=np.uint64))
        with pytest.raises(RuntimeError, match="record order"):
            allgather_minimizers(ctx, None, [bad], [0], 1)
        bad.free()
    finally:
        ctx.close()
