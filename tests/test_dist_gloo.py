"""The N>1 path on CPU: world_size-2 and -3 `gloo` process groups exercising the schedule of
tests/dist_double.py (bitwise-AND all-reduce = direct reduce-scatter + local AND + all-gather; all-gather(v)
of minimizer lists; genome->rank partition).  The AND operator is injected: on the GPU it is the HIP
kernel behind nts_and_raw, here a tensor op stands in so the communication pattern can run without a GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import dist_double as ndist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, nbytes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = ndist.padded_len(nbytes, world)
        rng = np.random.default_rng(100 + rank)
        mine = rng.integers(0, 256, size=n, dtype=np.uint8)
        mine[nbytes:] = 0
        buf = torch.from_numpy(mine.copy())

        def and_into(a, b):
            a.bitwise_and_(b)
        ndist.allreduce_and(buf, and_into)
        # minimizer lists of different lengths
        cnt = 5 + 3 * rank
        h1 = torch.arange(cnt, dtype=torch.int64) + 1000 * rank
        rec = torch.full((cnt,), rank, dtype=torch.int32)
        pos = torch.arange(cnt, dtype=torch.int64) * 7
        got = ndist.allgather_lists(h1, rec, pos, genome_id=10 + rank)
        q.put((rank, mine, buf.numpy().copy(), [(g, a.numpy().copy(), b.numpy().copy(), c.numpy().copy())
                                                 for g, a, b, c in got]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nbytes", [(2, 4096), (2, 1000), (3, 50_008)])
def test_and_allreduce_and_allgather(world, nbytes):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nbytes, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, mine, reduced, lists = q.get(timeout=120)
        res[rank] = (mine, reduced, lists)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = res[0][0].copy()
    for r in range(1, world):
        expect &= res[r][0]
    for r in range(world):
        assert np.array_equal(res[r][1], expect)            # every rank holds the AND of all filters
        lists = res[r][2]
        assert [g for g, *_ in lists] == [10 + i for i in range(world)]
        for i, (_, h1, rec, pos) in enumerate(lists):
            assert h1.tolist() == [1000 * i + j for j in range(5 + 3 * i)]
            assert rec.tolist() == [i] * (5 + 3 * i)
            assert pos.tolist() == [7 * j for j in range(5 + 3 * i)]


def test_partition_and_padding():
    assert ndist.genomes_of_rank(8, 3, 8) == [3]
    assert ndist.genomes_of_rank(3, 0, 2) == [0, 2] and ndist.genomes_of_rank(3, 1, 2) == [1]
    assert sorted(sum((ndist.genomes_of_rank(11, r, 4) for r in range(4)), [])) == list(range(11))
    for nbytes in (8, 1000, 493723632):
        for world in (1, 2, 4, 8):
            n = ndist.padded_len(nbytes, world)
            assert n >= nbytes and n % (16 * world) == 0 and n - nbytes < 16 * world


def _packed_worker(rank, world, port, q):
    import ctypes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_lists, cap = 2, 40
        pg = ndist.PackedListGather(n_lists, cap, "cpu")
        seen = []
        for step in range(3):                       # three steps over two buffer sets: the third reuses the first
            pg.begin()
            for i in range(n_lists):
                cnt = 3 + 5 * rank + 2 * i + step
                h1p, recp, posp = pg.slot_ptrs(i)   # what nts_mx_export would fill on the device
                h1 = np.arange(cnt, dtype=np.int64) + 1000 * rank + 100 * i + step
                rec = np.full(cnt, 10 * rank + i, dtype=np.int32)
                pos = np.arange(cnt, dtype=np.int64) * (7 + step)
                ctypes.memmove(h1p, h1.ctypes.data, h1.nbytes)
                ctypes.memmove(recp, rec.ctypes.data, rec.nbytes)
                ctypes.memmove(posp, pos.ctypes.data, pos.nbytes)
                pg.set_count(i, cnt, genome_id=rank * n_lists + i)
            t = pg.turn
            pg.post()
            pg.drain()
            seen.append([(g, a.numpy().copy(), b.numpy().copy(), c.numpy().copy()) for g, a, b, c in pg.lists_of(t)])
        q.put((rank, seen))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_packed_list_gather(world):
    "one all-gather per step carries every rank's lists (fixed slots + header), double-buffered"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_packed_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        for step in range(3):
            got = res[rank][step]
            assert [g for g, *_ in got] == list(range(2 * world))
            for r in range(world):
                for i in range(2):
                    g, h1, rec, pos = got[2 * r + i]
                    cnt = 3 + 5 * r + 2 * i + step
                    assert np.array_equal(h1, np.arange(cnt) + 1000 * r + 100 * i + step)
                    assert np.array_equal(rec, np.full(cnt, 10 * r + i)) and np.array_equal(pos, np.arange(cnt) * (7 + step))
