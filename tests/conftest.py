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

TESTS_DIR = os.path.dirname(os.path.abspath(__file__))
if TESTS_DIR not in sys.path:          # test modules share helpers (e.g. test_gpu_decode imports test_decode_oracle)
    sys.path.insert(0, TESTS_DIR)


def load_h256_fixture(name):
    """A fixture of tests/golden/make_golden_h256.py: returns (npz, categorical params, gaussian params).  The
    weights are regenerated from the recorded seeds with the oracle's initialiser and must hash to the recorded
    SHA-256 (they were loaded into the imported reference with strict=True when the fixture was made)."""
    import numpy as np
    from oracle import difusco_oracle as O
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    H, L = int(z["hidden"]), int(z["n_layers"])
    p_cat = O.init_params(H, L, 2, seed=int(z["seed_cat"]))
    p_gau = O.init_params(H, L, 1, seed=int(z["seed_gau"]))
    assert O.params_sha256(p_cat) == str(z["sha_cat"]) and O.params_sha256(p_gau) == str(z["sha_gau"]), \
        "regenerated weights differ from the ones the fixture was made with (torch RNG changed?)"
    return z, p_cat, p_gau
