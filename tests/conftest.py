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
    # The built libraries are git-ignored.  In a fresh checkout on a box with the toolchain
    # (the CPU container), build them once before collection; on the GPU box the prebuilt
    # files travel with the tree and nothing is built here.
    import shutil
    import subprocess
    hip_so = os.path.join(ROOT, "map_neo_amd", "libmdd_hip.so")
    if not os.path.exists(hip_so) and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "map_neo_amd", "csrc"), "all"])
    ref_so = os.path.join(ROOT, "oracle", "libmdd_ref.so")
    if not os.path.exists(ref_so) and shutil.which("gcc"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libmdd_ref.so"])


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
