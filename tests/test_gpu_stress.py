"""Short randomised parity runs (scripts/stress_parity.py, scripts/stress_pipeline.py) as part of the GPU suite: a few
seconds each here; the long runs are recorded in profiles/README.md."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "scripts", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed", [11, 12])
def test_random_sketches_and_filters_match_the_oracle(seed, capsys):
    _load("stress_parity").main(["--seconds", "6", "--seed", str(seed)])     # sys.exit(1) on a mismatch
    assert capsys.readouterr().out.startswith("ok:")


def test_random_families_end_to_end_match_the_oracle(capsys):
    _load("stress_pipeline").main(["--seconds", "12", "--seed", "7"])
    # (families of many distant genomes may share no chain of four minimizers: both sides then say "no paths found" before the summary line)
    assert capsys.readouterr().out.strip().splitlines()[-1].startswith("ok:")
