import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture()
def in_tmp_cwd(tmp_path):
    "the pipeline writes its artefacts into the CWD, like the reference does"
    old = os.getcwd()
    os.chdir(tmp_path)
    try:
        yield tmp_path
    finally:
        os.chdir(old)


@pytest.fixture(scope="session")
def ctx_x():
    """A context on the EXPERIMENTS build of the library (ntsynt_amd/libntsynt_hip_exp.so, csrc/nts_knobs.h): the tests that pick a kernel
    variant, force a fallback or cut a list short through an environment switch run on it; everything else runs on the product build,
    which does not contain those switches."""
    from ntsynt_amd.device import Context
    c = Context(0, variant="experiments")
    yield c
    c.close()
