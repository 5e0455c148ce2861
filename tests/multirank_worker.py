"""One rank of tests/test_gpu_multirank.py's exchange tests: drives the product's two exchanges (nts_bf_allreduce_and,
nts_mx_allgather in libntsynt_hip.so) between `world` processes that share GPU 0, over the librccl stand-in named by
NTS_RCCL_LIB.  No torch here: the communicator id travels through a file, as any launcher could do it.

    python multirank_worker.py RANK WORLD ID_FILE CASE

Prints one JSON line; exit status 0 = every check held on this rank."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def exchange_through(path):
    def exchange(ident):
        if ident is not None:
            with open(path + ".tmp", "wb") as fh:
                fh.write(ident)
            os.rename(path + ".tmp", path)
            return ident
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > 120:
                raise RuntimeError("the communicator id never arrived")
            time.sleep(0.01)
        with open(path, "rb") as fh:
            return fh.read()
    return exchange


def filter_bytes(rank, nbytes, density):
    "seeded filter contents of a rank: every rank can compute every other rank's"
    rng = np.random.default_rng(1000 + rank)
    bits = rng.random(nbytes * 8) < density
    return np.packbits(bits, bitorder="little")


def list_of(g, n):
    rng = np.random.default_rng(77 + g)
    h1 = rng.integers(0, 2**64, size=n, dtype=np.uint64)
    rec = np.sort(rng.integers(0, 5, size=n)).astype(np.uint32)
    pos = rng.integers(0, 2**40, size=n, dtype=np.uint64)
    return h1, rec, pos


def main():
    rank, world, id_file, case = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    from ntsynt_amd.device import BloomFilter, Comm, Context, Minimizers
    ctx = Context(0)
    comm = Comm(ctx, world, rank, exchange_through(id_file))
    out = {"rank": rank, "world": world, "case": case, "library": ctx.lib.nts_comm_library().decode()}
    assert comm.world == world and comm.rank == rank
    assert ctx.lib.nts_comm_world(comm.h) == world and ctx.lib.nts_comm_rank(comm.h) == rank
    if case == "allreduce":
        # sizes that are not a multiple of 16 x world (the chunks get padding), a one-chunk-per-piece and a many-pieces run
        checked = []
        for nbytes in (8, 1000, 123_456 + 8 * world, 3_000_008):
            mine = filter_bytes(rank, nbytes, 0.6)
            bf = BloomFilter(ctx, nbytes, 24, world=world)
            bf.from_numpy(mine)
            comm.allreduce_and(bf)
            expect = mine.copy()
            for r in range(world):
                expect &= filter_bytes(r, nbytes, 0.6)
            got = bf.to_numpy()
            assert got.size == nbytes and np.array_equal(got, expect), f"AND all-reduce differs at {nbytes} bytes"
            assert abs(bf.get_fpr() - np.unpackbits(expect).sum() / (nbytes * 8)) < 1e-12      # popcount sees the new bits
            checked.append(nbytes)
            bf.free()
        # a rank that owns no genome contributes the identity
        bf = BloomFilter(ctx, 5000, 24, world=world, ones=(rank == world - 1))
        if rank != world - 1:
            bf.from_numpy(filter_bytes(rank, 5000, 0.5))
        comm.allreduce_and(bf)
        expect = np.full(5000, 0xFF, dtype=np.uint8)
        for r in range(world - 1):
            expect &= filter_bytes(r, 5000, 0.5)
        assert np.array_equal(bf.to_numpy(), expect), "identity rank"
        bf.free()
        out["sizes"] = checked
    elif case == "groups":
        # fewer genomes than ranks: rank r holds the filter of a shard of genome r mod G; the common filter is the AND over genomes of
        # the OR over each genome's shards.  Dense filters (chunks gathered) and all-but-empty ones (set-bit indices gathered).
        G = 2
        group_of = [r % G for r in range(world)]
        modes = []
        for nbytes, density in ((1000, 0.5), (123_456 + 8 * world, 0.4), (3_000_008, 0.3), (3_000_008, 0.0008), (64, 0.0)):
            mine = filter_bytes(rank, nbytes, density)
            bf = BloomFilter(ctx, nbytes, 24, world=world)
            bf.from_numpy(mine)
            comm.allreduce_groups(bf, group_of)
            expect = np.full(nbytes, 0xFF, dtype=np.uint8)
            for g in range(G):
                union = np.zeros(nbytes, dtype=np.uint8)
                for r in range(world):
                    if group_of[r] == g:
                        union |= filter_bytes(r, nbytes, density)
                expect &= union
            got = bf.to_numpy()
            assert got.size == nbytes and np.array_equal(got, expect), f"grouped all-reduce differs at {nbytes} bytes, density {density}"
            assert bf.popcount() == int(np.unpackbits(expect).sum())
            modes.append(bool(comm.last_sparse()))
            bf.free()
        out["sparse"] = modes
        # a group without a rank is refused (its OR would clear the filter)
        bf = BloomFilter(ctx, 1000, 24, world=world)
        try:
            comm.allreduce_groups(bf, [0] * (world - 1) + [2])
            raise AssertionError("a group without a rank was accepted")
        except Exception as exc:                  # noqa: BLE001
            if isinstance(exc, AssertionError):
                raise
            out["rejects_empty_group"] = True
        bf.free()
    elif case == "parts":
        # a family's records shared out by bases across genome boundaries (pipeline.partition_plan): a rank holds a filter per genome its
        # range touches -- two here where a range crosses a boundary, none for a rank without records; lists likewise, several per rank
        plans = {2: [[0, 1], [1, 2]], 3: [[0], [0, 1], [1]], 4: [[0, 1], [1], [1, 2], []]}
        slot_group = plans[world]
        n_slots = max(1, max(len(x) for x in slot_group))
        n_groups = 1 + max(g for row in slot_group for g in row)
        modes = []
        for nbytes, density in ((1000, 0.5), (123_456 + 8 * world, 0.4), (3_000_008, 0.3), (3_000_008, 0.0008)):
            mine = [BloomFilter(ctx, nbytes, 24, world=world) for _ in range(max(1, len(slot_group[rank])))]
            for s_, f in enumerate(mine[:len(slot_group[rank])]):
                f.from_numpy(filter_bytes(10 * rank + s_, nbytes, density))
            comm.allreduce_parts(mine, slot_group, n_slots, n_groups)
            expect = np.full(nbytes, 0xFF, dtype=np.uint8)
            for g in range(n_groups):
                union = np.zeros(nbytes, dtype=np.uint8)
                for r, row in enumerate(slot_group):
                    for s_, gg in enumerate(row):
                        if gg == g:
                            union |= filter_bytes(10 * r + s_, nbytes, density)
                expect &= union
            got = mine[0].to_numpy()
            assert got.size == nbytes and np.array_equal(got, expect), f"all-reduce over parts differs at {nbytes} bytes, density {density}"
            assert mine[0].popcount() == int(np.unpackbits(expect).sum())
            modes.append(bool(comm.last_sparse()))
            for f in mine:
                f.free()
        out["sparse"] = modes
        # a plan whose contributions are not a rank's first slots is refused, before anything is sent
        bf = BloomFilter(ctx, 1000, 24, world=world)
        try:
            comm.allreduce_parts([bf], [[-1, 0]] + [[0, -1]] * (world - 1), 2, 1)
            raise AssertionError("a gap in a rank's slots was accepted")
        except Exception as exc:                  # noqa: BLE001
            if isinstance(exc, AssertionError):
                raise
            out["rejects_gaps"] = True
        bf.free()
        # exchange 2 with several lists per rank: list numbers in family order, the slot count said by the caller
        n_lists = sum(len(x) for x in slot_group)
        base = sum(len(x) for x in slot_group[:rank])
        ids = [base + i for i in range(len(slot_group[rank]))]
        sizes = [37 * (g + 1) % 101 for g in range(n_lists)]
        local = [Minimizers.from_numpy(ctx, *list_of(g, sizes[g])) for g in ids]
        everything = comm.allgather_minimizers(local, ids, n_lists, n_slots)
        for g, mx in enumerate(everything):
            h1, rec, pos = mx.to_numpy()
            eh, er, ep = list_of(g, sizes[g])
            assert np.array_equal(h1, eh) and np.array_equal(rec, er) and np.array_equal(pos, ep), f"list {g} of {n_lists}"
            mx.free()
        for mx in local:
            mx.free()
    elif case == "allgather":
        # n_total lists, genome g on rank g mod world: uneven shares, an empty list, a rank that holds fewer than the others
        for n_total, sizes in ((5, [1000, 0, 37, 4099, 1]), (world, [3] * world), (2 * world + 1, [11 * (g + 1) for g in range(2 * world + 1)])):
            ids = [g for g in range(n_total) if g % world == rank]
            local = [Minimizers.from_numpy(ctx, *list_of(g, sizes[g])) for g in ids]
            everything = comm.allgather_minimizers(local, ids, n_total)
            assert len(everything) == n_total
            for g, mx in enumerate(everything):
                h1, rec, pos = mx.to_numpy()
                eh, er, ep = list_of(g, sizes[g])
                assert np.array_equal(h1, eh) and np.array_equal(rec, er) and np.array_equal(pos, ep), f"list {g} of {n_total}"
                mx.free()
            for mx in local:                      # the rank's own lists are copied, not aliased: still intact
                mx.free()
        # bad call: the same genome number from two ranks
        try:
            dup = comm.allgather_minimizers([Minimizers.from_numpy(ctx, *list_of(0, 5))], [0], world)
            raise AssertionError("repeated genome numbers were accepted")
        except Exception as exc:                  # noqa: BLE001
            if isinstance(exc, AssertionError):
                raise
            out["rejects_repeats"] = True
    else:
        raise SystemExit("unknown case " + case)
    comm.close()
    ctx.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
