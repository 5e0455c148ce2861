"""A launch counts its work-items in 32 bits per dimension: gridDim.x * blockDim.x beyond 2^32 - 1 runs truncated and HIP reports nothing.
Found with a selection forced to list next to nothing (nts_sketch_mode 2, c = 1, w = 16) on 600 Mbp: 8.9 M uncovered ranges, one window
tile of 512 lanes each -- every tile past the 2^23rd was dropped and a third of the minimizers went missing without an error.  The window
kernel now goes out in as many launches as it takes, and every launch of the library refuses to run truncated (NTS_LAUNCH, nts_internal.h).
Reference semantics: the list is btllib indexlr's whatever way it is computed (bin/ntsynt_run_pipeline.smk:74-85).  Needs an MI355X."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _lists(ctx, g, k, w, mode, c=0):
    from ntsynt_amd.device import sketch
    ctx.sketch_mode(mode, c)
    ctx.sketch_tiers("never")
    try:
        mx = sketch(ctx, g, k, w, None)
        out = mx.to_numpy()
        mx.free()
        return out
    finally:
        ctx.sketch_mode("auto")
        ctx.sketch_tiers("auto")


def test_more_uncovered_ranges_than_one_launch_holds():
    from ntsynt_amd.device import Context, Genome
    ctx = Context(0)
    try:
        g = Genome.synth(ctx, 620_000_000, 24, 20240207, 2, 0.005)
        dense = _lists(ctx, g, 24, 16, "dense")
        pruned = _lists(ctx, g, 24, 16, "pruned", 1)
        ranges = ctx.sketch_stats()[1]
        assert ranges > (1 << 23), "the case no longer has more window tiles than 2^32 / 512"
        assert len(dense[0]) == len(pruned[0])
        for a, b in zip(dense, pruned):
            assert np.array_equal(a, b)
        g.free()
    finally:
        ctx.close()


def test_a_launch_that_would_run_truncated_is_refused(monkeypatch):
    "experiments build: the window tiles of the same case in ONE launch -- the call must end in an error, not in a shorter list"
    from ntsynt_amd.device import Context, Genome, NtsError
    monkeypatch.setenv("NTS_WIN_TILES_PER_LAUNCH", str(1 << 30))
    ctx = Context(0, variant="experiments")
    try:
        g = Genome.synth(ctx, 620_000_000, 24, 20240207, 2, 0.005)
        with pytest.raises(NtsError, match="2\\^32"):
            _lists(ctx, g, 24, 16, "pruned", 1)
        # the context is usable afterwards
        monkeypatch.delenv("NTS_WIN_TILES_PER_LAUNCH")
        small = Genome.synth(ctx, 2_000_000, 24, 7, 1, 0.005)
        a = _lists(ctx, small, 24, 16, "dense")
        b = _lists(ctx, small, 24, 16, "pruned", 1)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        small.free()
        g.free()
    finally:
        ctx.close()
