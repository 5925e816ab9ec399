import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # the HIP library is a build product (git-ignored): a fresh checkout compiles it once, here (hipcc cross-compiles
    # gfx950 without a GPU); nothing is rebuilt when it exists
    if not os.path.exists(os.path.join(REPO, "vrdone_amd", "csrc", "libvrdone_hip.so")):
        import __graft_entry__
        __graft_entry__.build()


def load_case(name):
    """(model_config, inference_config, ordered [(key, shape)]) of a golden case."""
    with open(os.path.join(GOLDEN, f"state_keys_{name}.json")) as f:
        d = json.load(f)
    return d["model_config"], d["inference_config"], [(k, tuple(s)) for k, s in d["keys"]]


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
