#!/usr/bin/env python3
"""The reference's first published row as a shape (two 3 Gbp genomes at 0.1 %, -d 0.1: refinement rounds at w = 100 and w = 10) end to
end, with the time line and the engine's step times: where the short-window rounds' time goes.  NTS_LIB_VARIANT=experiments
NTS_WIN_FUSE=0 in the environment: the same run through the key-array path."""
import argparse
import json
import os
import shutil
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    args = argparse.Namespace(family="structural", substitutions_only=False, k=24, w=1000, fpr=0.025)
    work = tempfile.mkdtemp(prefix="nts_d01_", dir=os.environ.get("TMPDIR", "/tmp"))
    os.environ["NTS_ENGINE_TIMES"] = "1"
    try:
        paths = bench.e2e_inputs(args, 0, 2, 3_000_000_000, 24, 0.001, work)
        for rep in range(2):
            sub = os.path.join(work, f"run{rep}")
            os.makedirs(sub)
            r = bench.e2e_leg(args, 0, 2, 3_000_000_000, 24, 0.001, sub, paths=paths)
            print(json.dumps({k_: r.get(k_) for k_ in ("seconds", "time_line_s", "stages_s", "engine_times_s", "allocator", "blocks", "oracle_checked")}), flush=True)
            shutil.rmtree(sub, ignore_errors=True)
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
