"""The library's device allocation cache (csrc/ntsynt_hip.hip, namespace nts_mem): freed allocations are kept and handed out again whole
or in pieces; pieces that come back are joined with their free neighbours; nts_mem_trim gives wholly free allocations back to the
driver.  Observed through filters of chosen sizes (BloomFilter = one allocation of its byte count), nts_alloc_stats (calls of hipMalloc /
hipFree made by the library) and nts_mem_cache_stats."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MB = 1 << 20


@pytest.fixture(scope="module")
def ctx():
    from ntsynt_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def _calls(ctx):
    return ctx.alloc_stats()[0]


def test_freed_allocation_is_cut_into_pieces_and_joined_again(ctx):
    from ntsynt_amd.device import BloomFilter
    ctx.sync()
    ctx.mem_trim()                                            # (start from an empty cache: what other tests left is gone)
    assert ctx.mem_cache_stats()[0] == 0
    big = BloomFilter(ctx, 64 * MB, 24)
    c0 = _calls(ctx)
    big.free()                                                # kept: no hipFree
    assert _calls(ctx) == c0 and ctx.mem_cache_stats()[0] == 64 * MB
    hits0 = ctx.mem_cache_stats()[1]
    parts = [BloomFilter(ctx, 8 * MB, 24) for _ in range(4)]  # four pieces of the kept allocation: no hipMalloc
    assert _calls(ctx) == c0 and ctx.mem_cache_stats() == (32 * MB, hits0 + 4)
    # the pieces are separate memory: each filter keeps its own bits
    for i, f in enumerate(parts):
        f.from_numpy(np.full(8 * MB, i + 1, dtype=np.uint8))
    for i, f in enumerate(parts):
        got = f.to_numpy()
        assert got[0] == i + 1 and got[-1] == i + 1 and int(got.sum(dtype=np.uint64)) == (i + 1) * 8 * MB
    # freed out of order, they join: the whole allocation is one range again and serves a request of its full size
    for i in (1, 3, 0, 2):
        parts[i].free()
    assert ctx.mem_cache_stats()[0] == 64 * MB and _calls(ctx) == c0
    again = BloomFilter(ctx, 64 * MB, 24)
    assert _calls(ctx) == c0 and ctx.mem_cache_stats()[0] == 0
    # a request larger than anything kept goes to the driver; a partly used allocation stays when the cache is trimmed
    half = None
    again.free()
    half = BloomFilter(ctx, 40 * MB, 24)                      # piece of the 64 MB allocation
    assert ctx.mem_cache_stats()[0] == 24 * MB
    bigger = BloomFilter(ctx, 100 * MB, 24)
    assert _calls(ctx) == c0 + 1
    assert ctx.mem_trim() == 0 and ctx.mem_cache_stats()[0] == 24 * MB          # (nothing wholly free)
    half.free()
    assert ctx.mem_cache_stats()[0] == 64 * MB
    assert ctx.mem_trim() == 64 * MB and ctx.mem_cache_stats()[0] == 0 and _calls(ctx) == c0 + 2
    bigger.free()
    assert ctx.mem_trim() >= 100 * MB


def test_small_requests_and_odd_sizes(ctx):
    "requests below 64 KiB are not cut from kept large allocations; sizes that are no multiple of the grain are rounded up and still join"
    from ntsynt_amd.device import BloomFilter
    ctx.sync()
    ctx.mem_trim()
    a = BloomFilter(ctx, 3 * MB + 8, 24)
    b = BloomFilter(ctx, 1024, 24)
    a.free()
    b.free()
    kept = ctx.mem_cache_stats()[0]
    assert 3 * MB + 8 <= kept < 3 * MB + 8 + 8192             # (the small one is not kept)
    c = BloomFilter(ctx, MB + 8, 24)
    d = BloomFilter(ctx, MB + 16, 24)
    c.free()
    d.free()
    assert ctx.mem_cache_stats()[0] == kept
    ctx.mem_trim()


def test_small_requests_live_in_a_slab_of_their_own(ctx):
    """below 64 KiB: cut from an 8 MB slab kept for small requests (at most one driver call, when there is none yet or it is full); a kept
    large allocation is left whole; pieces are separate memory and go back where they came from"""
    from ntsynt_amd.device import BloomFilter
    ctx.sync()
    ctx.mem_trim()
    a = BloomFilter(ctx, 2 * MB, 24)
    a.free()
    c0 = _calls(ctx)
    small = [BloomFilter(ctx, 8 * (i + 1), 24) for i in range(5)]
    assert _calls(ctx) <= c0 + 1 and ctx.mem_cache_stats()[0] == 2 * MB
    c1 = _calls(ctx)
    for i, f in enumerate(small):
        f.from_numpy(np.full(8 * (i + 1), 7 * i + 1, dtype=np.uint8))
    for i, f in enumerate(small):
        assert set(f.to_numpy().tolist()) == {7 * i + 1}
    for f in small:
        f.free()
    more = [BloomFilter(ctx, 1000 + i, 24) for i in range(200)]              # the slab serves these too: no driver call
    for f in more:
        f.free()
    assert ctx.mem_cache_stats()[0] == 2 * MB and _calls(ctx) == c1
    assert ctx.mem_trim() == 2 * MB


def test_reserve_serves_later_allocations_without_the_driver(ctx):
    """nts_mem_reserve: one driver allocation that the following requests are cut from -- none of them reaches the driver; the reserved
    allocation is not counted against the cache's limit; nts_mem_events tells what a stage asked of the driver"""
    from ntsynt_amd.device import BloomFilter
    ctx.sync()
    ctx.mem_trim()
    ev0 = ctx.mem_events()
    got = ctx.mem_reserve(256 * MB)
    assert got == 256 * MB and ctx.mem_cache_stats()[0] == 256 * MB
    d = ctx.mem_events_since(ev0)
    assert d["reserve_calls"] == 1 and d["hipMalloc_hipFree_calls"] >= 1 and abs(d["GB_from_driver"] - 0.268) < 0.02
    ev1 = ctx.mem_events()
    fs = [BloomFilter(ctx, 40 * MB, 24) for _ in range(6)]                    # 240 MB of the 256
    for i, f in enumerate(fs):
        f.from_numpy(np.full(40 * MB, i + 7, dtype=np.uint8))
    assert all(int(f.to_numpy()[123]) == i + 7 for i, f in enumerate(fs))
    small = [BloomFilter(ctx, 4096, 24) for _ in range(20)]                   # small requests: their own slab, not the reserved one
    for f in fs + small:
        f.free()
    d = ctx.mem_events_since(ev1)
    assert d["hipMalloc_hipFree_calls"] == 0 and d["allocations_served_from_kept_memory"] >= 26 and d["GB_to_driver"] == 0
    assert ctx.mem_cache_stats()[0] == 256 * MB                               # joined again: one free range
    whole = BloomFilter(ctx, 256 * MB, 24)
    assert ctx.mem_events_since(ev1)["hipMalloc_hipFree_calls"] == 0
    whole.free()
    assert ctx.mem_trim() == 256 * MB                                         # a trim takes the reserved allocation too


def test_a_small_block_does_not_hold_a_large_kept_allocation(ctx):
    "ADVICE r5: a long-lived 4 KB workspace used to be cut from a multi-GB kept block, which then could not go back to the driver"
    from ntsynt_amd.device import BloomFilter
    ctx.sync()
    ctx.mem_trim()
    big = BloomFilter(ctx, 128 * MB, 24)
    big.free()                                                                # kept
    tiny = BloomFilter(ctx, 4096, 24)                                         # lives on
    assert ctx.mem_cache_stats()[0] == 128 * MB                               # not cut from the kept block
    assert ctx.mem_trim() == 128 * MB                                         # which is wholly free and goes
    tiny.free()


def test_a_pipeline_run_reserves_its_plan_once_and_cuts_everything_from_it(tmp_path):
    """ntsynt_amd.pipeline.run: one nts_mem_reserve sized from the file sizes (plan_bytes) before the first file is read; the filter, the
    Bloom build's workspaces, the genomes and the sketch / graph workspaces are cut from it -- next to the reserve the run makes a
    handful of driver calls at most (pinned slabs of the small requests, what the ingest asked while the reserve was under way)"""
    import os
    from ntsynt_amd import _lib, pipeline, synth
    from ntsynt_amd.device import mem_events, mem_events_since
    paths = synth.make_family(str(tmp_path), 3, 60_000_000, 3, 0.01, seed=77)
    sizes = sorted(os.path.getsize(p) for p in sorted(paths))
    plan = pipeline.plan_bytes([os.path.getsize(p) for p in sorted(paths)], 0.025, True)
    assert plan > 2 * (1 << 30) + 1.3 * sum(sizes)
    lib = _lib.load()
    lib.nts_mem_trim()
    ev0 = mem_events(lib)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        eng = pipeline.run(paths, k=24, w=1000, prefix="r", log=lambda *a: None)
    finally:
        os.chdir(cwd)
    d = mem_events_since(lib, ev0)
    assert eng.reserved_bytes >= plan - (1 << 20) and d["reserve_calls"] == 1
    assert d["allocations_retried_after_emptying_the_cache"] == 0
    assert d["allocations_served_from_kept_memory"] > 50
    # the reserve, the small requests' slab, and at most a few allocations of the first file's ingest that ran next to the reserve
    assert d["hipMalloc_hipFree_calls"] <= 12, d
    assert d["GB_from_driver"] < plan / 1e9 + 1.0
    assert len(eng.outputs["r.synteny_blocks.tsv"].splitlines()) > 10
    assert "memory_reserved" in dict(eng.stage_marks)
