"""ntsynt_amd.pipeline.run under torch.distributed (gloo, world size 2 and 3) on CPU.

What runs here is the product's orchestration -- genome->rank ownership, per-rank filter AND + bitwise-AND
all-reduce, owner-broadcast of minimizer lists, replicated graph stage, rank-0 output -- against a test-double
backend (oracle sketch, numpy graph build, CPU tensors), because the GPU backend cannot run without a GPU.
The result must be byte-identical to the single-process oracle pipeline."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ntsynt_amd import synth


class OracleBackend:
    "test double with GpuBackend's interface"

    def __init__(self):
        from oracle import nts_oracle as O
        self.O = O

    def load_genome(self, path):
        from ntsynt_amd import fasta as fa
        g = self.O.read_fasta(path)
        g.recs = fa.read_fasta(path)
        return g

    class _BF:
        pass

    def bf_new(self, nbytes, k, world=1, ones=False):
        from tests.dist_double import padded_len
        bf = self._BF()
        bf.k, bf.nbytes = k, nbytes
        n = padded_len(nbytes, world) if world > 1 else nbytes
        bf.tensor = torch.zeros(n, dtype=torch.uint8)
        if ones:
            bf.tensor[:nbytes] = 0xFF
        return bf

    def _arr(self, bf):
        return bf.tensor.numpy()[:bf.nbytes]

    def bf_insert(self, bf, genome):
        a = self._arr(bf)
        a |= self.O.bf_build(genome, bf.k, bf.nbytes)

    def bf_and(self, acc, other):
        a = self._arr(acc)
        a &= self._arr(other)

    def bf_clear(self, bf):
        bf.tensor.zero_()

    def bf_fpr(self, bf):
        return self.O.bf_fpr(np.ascontiguousarray(self._arr(bf)))

    def bf_bits(self, bf):
        return np.ascontiguousarray(self._arr(bf)).copy()

    def and_into(self, a, b):
        a.bitwise_and_(b)

    def allreduce_and(self, bf):
        "exchange 1 of the test double: the schedule of tests/dist_double.py over gloo, CPU tensors"
        from tests.dist_double import allreduce_and
        allreduce_and(bf.tensor, self.and_into)

    def sync(self):
        pass

    def sketch(self, genome, k, w, bf, masks=None):
        from tests.helpers import oracle_flat
        g = genome
        if masks:
            seqs = []
            for r in range(len(g.names)):
                buf = bytearray(g.record(r))
                for mr, s, e in masks:
                    if mr == r:
                        s, e = max(0, s), min(len(buf), e)
                        if e > s:
                            buf[s:e] = b"N" * (e - s)
                seqs.append(bytes(buf))
            g = self.O.Genome(g.names, seqs)
        bits = None if bf is None else np.ascontiguousarray(self._arr(bf))
        return oracle_flat(self.O.minimize(g, k, w, bits))

    def graph(self, lists, keeps, list_ids):
        from tests.graph_ref import build_graph_numpy
        return build_graph_numpy(lists, keeps, list_ids)

    def walk(self, nv, eu, ev):
        from ntsynt_amd.graph import walk_chains
        return walk_chains(nv, eu, ev)

    def to_comm(self, arr, dtype):
        return torch.from_numpy(np.ascontiguousarray(arr).view(dtype).copy())

    def comm_empty(self, n, dtype):
        return torch.empty(n, dtype=dtype)

    def close(self):
        pass


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


KW = dict(k=24, w=400, w_rounds=[100, 10], indel=500, merge=3000, block_size=300)


def _worker(rank, world, port, paths, workdir, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ntsynt_amd import pipeline
        os.chdir(workdir)
        eng = pipeline.run(paths, prefix="d", backend=OracleBackend(), log=lambda *a: None, **KW)
        q.put((rank, eng.outputs, sorted(os.listdir(workdir))))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_genomes", [(2, 3), (3, 2)])
def test_distributed_pipeline_matches_single_process(tmp_path, world, n_genomes):
    from oracle import synteny_oracle as SO
    paths = synth.make_family(str(tmp_path), n_genomes, 700_000, 2, 0.01, seed=17, micro=6)
    ref_dir = tmp_path / "ref"
    os.makedirs(ref_dir)
    cwd = os.getcwd()
    try:
        os.chdir(ref_dir)
        ora = SO.run_pipeline(paths, prefix="d", **KW)
    finally:
        os.chdir(cwd)
    work = tmp_path / "dist"
    os.makedirs(work)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, paths, str(work), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, outputs, files = q.get(timeout=300)
        res[rank] = (outputs, files)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):                       # the replicated graph stage agrees on every rank
        for name in ("d.synteny_blocks.tsv", "d.pre-collinear-merge.synteny_blocks.tsv"):
            assert res[r][0][name] == ora.outputs[name], (r, name)
    # rank 0 left the reference's artefacts in the CWD, the other ranks nothing
    files = set(os.listdir(work))
    assert {"d.synteny_blocks.tsv", "d.pre-collinear-merge.synteny_blocks.tsv", "d.common.bf"} <= files
    assert not any(f.startswith(".ntsynt_rank") for f in files)
    for p in paths:
        assert f"{os.path.basename(p)}.fai" in files and f"{os.path.basename(p)}.k24.w400.tsv" in files
    assert open(work / "d.synteny_blocks.tsv").read() == ora.outputs["d.synteny_blocks.tsv"]
    from ntsynt_amd.pipeline import read_bf
    bits, _ = read_bf(str(work / "d.common.bf"))
    assert np.array_equal(bits, ora.bf)


def test_a_run_without_paths_keeps_the_finished_stages_outputs(tmp_path, monkeypatch, capsys):
    """Stage 3's "no paths found" (bin/ntsynt_synteny.py:630-632, exit 1) under the reference's Snakemake fails rule ntsynt_synteny
    only: <prefix>.common.bf (rule make_common_bf, smk:55-62), the minimizer TSVs (rule indexlr, smk:74-85) and the .fai files stay,
    no block table appears.  Same here, and the TSVs are the oracle pipeline's."""
    from ntsynt_amd import pipeline
    from oracle import synteny_oracle as SO
    paths = synth.make_family(str(tmp_path), 2, 300_000, 2, 0.35, seed=5)
    names = [f"{os.path.basename(p)}.k24.w400.tsv" for p in paths]
    texts = {}
    for side in ("ora", "own"):
        os.makedirs(tmp_path / side)
        monkeypatch.chdir(tmp_path / side)
        with pytest.raises(SystemExit) as e:
            if side == "ora":
                SO.run_pipeline(paths, prefix="d", **KW)
            else:
                pipeline.run(paths, prefix="d", backend=OracleBackend(), log=lambda *a: None, **KW)
        assert e.value.code == 1
        files = set(os.listdir("."))
        assert not any(f.endswith("synteny_blocks.tsv") for f in files)
        assert set(names) <= files
        texts[side] = [open(n).read() for n in names]
        if side == "own":
            assert "d.common.bf" in files and all(f"{os.path.basename(p)}.fai" in files for p in paths)
    assert texts["ora"] == texts["own"]
    assert "no paths found" in capsys.readouterr().out
