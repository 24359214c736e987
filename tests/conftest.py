import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def mini_docs():
    with open(os.path.join(GOLDEN, "mini_corpus.json"), encoding="utf8") as f:
        return json.load(f)


@pytest.fixture(scope="session")
def mini_expected():
    return dict(np.load(os.path.join(GOLDEN, "mini_expected.npz")))


@pytest.fixture(scope="session")
def golden_perms():
    return np.load(os.path.join(GOLDEN, "perms_seed1_K256.npy"))


@pytest.fixture(scope="session")
def engine():
    """HIP engine on cuda:0 (gpu tests only).  Fails loudly if the extension is missing."""
    from map_neo_amd.minhash import Engine
    return Engine(0)
