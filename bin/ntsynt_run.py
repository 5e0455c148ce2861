#!/usr/bin/env python3
"""ntsynt_run.py of the MI355X-native build: the reference's stage executable of that name, same command line (ntsynt_amd/stage_cli.py)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.realpath(__file__)), ".."))
from ntsynt_amd.stage_cli import run as main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
