"""Two ranks on one box: bench.py's and the pipeline's multi-rank code paths, end to end on the GPU, with gloo and host
copies standing in for RCCL (a 1-GPU box cannot host two RCCL ranks).  What RCCL itself adds -- the collectives on
device buffers -- is the part exercised by the driver's multi-GPU run; the schedules are covered on CPU by
tests/test_dist_gloo.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(n, script_args, env_extra, cwd):
    env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1", **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=600)


def test_bench_two_ranks(tmp_path):
    r = _torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "c4", "--genomes", "4",
                      "--mbp", "6", "--contigs", "2"], {"NTS_BENCH_BACKEND": "gloo"}, tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0 and out["steps"] == 2
    assert out["config"]["genomes_on_rank0"] == 2 and "c4: 4 synthetic" in out["config"]["workload"]
    assert out["bloom"]["allreduce_and_s"] > 0
    assert "e2e" not in out and "cpu_baseline" not in out          # single-GPU legs only


def test_pipeline_two_ranks_matches_single_rank(tmp_path):
    "bin/ntSynt under torchrun with two ranks (genomes sharded, AND all-reduce, list broadcasts) == one rank, byte for byte"
    from ntsynt_amd import synth
    paths = synth.make_family(str(tmp_path), 3, 1_500_000, 2, 0.01, seed=44, micro=6)
    one = tmp_path / "one"
    two = tmp_path / "two"
    os.makedirs(one)
    os.makedirs(two)
    args = ["-k", "24", "-w", "500", "-d", "1", "--prefix", "p", "--indel", "5000", "--merge", "20000", "--force"] + paths
    env = dict(os.environ, PYTHONPATH=ROOT)
    subprocess.run([sys.executable, os.path.join(ROOT, "bin", "ntSynt")] + args, cwd=one, env=env, check=True,
                   stdout=subprocess.DEVNULL, timeout=600)
    r = _torchrun(2, [os.path.join(ROOT, "bin", "ntSynt")] + args, {"NTS_DIST_BACKEND": "gloo"}, two)
    assert r.returncode == 0, r.stderr[-3000:]
    for name in ("p.synteny_blocks.tsv", "p.pre-collinear-merge.synteny_blocks.tsv", "p.common.bf"):
        assert (one / name).read_bytes() == (two / name).read_bytes(), name
    assert len((one / "p.synteny_blocks.tsv").read_text().splitlines()) >= 6
