import glob
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_cases():
    return sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "case_*.npz")))


@pytest.fixture(scope="session")
def load_golden():
    cache = {}

    def _load(name):
        if name not in cache:
            with np.load(os.path.join(GOLDEN_DIR, f"case_{name}.npz")) as z:
                cache[name] = {k: z[k] for k in z.files}
        return cache[name]

    return _load


@pytest.fixture(scope="session")
def h10_fci():
    """H10 / STO-3G / 5 FCI training states with physical integrals (tests/golden/make_h10_fci.py)."""
    with np.load(os.path.join(GOLDEN_DIR, "h10_fci_t5.npz")) as z:
        return {k: z[k] for k in z.files}


def bundle_from_golden(g):
    """oracle.AOBundle from a golden dict (tests only)."""
    from oracle.evcont_oracle import AOBundle
    return AOBundle(S=g["S"], hcore=g["hcore"], eri=g["eri"], ipovlp=g["ipovlp"], dhcore=g["dhcore"],
                    eri_ip1=g["eri_ip1"], aoslices=g["aoslices"], enuc=float(g["enuc"]), gnuc=g["gnuc"])


def ao_from_golden(g):
    """evcont_amd.synthetic.AOArrays from a golden dict."""
    from evcont_amd.synthetic import AOArrays
    return AOArrays(S=g["S"], hcore=g["hcore"], eri=g["eri"], ipovlp=g["ipovlp"], dhcore=g["dhcore"],
                    eri_ip1=g["eri_ip1"], aoslices=g["aoslices"], enuc=float(g["enuc"]), gnuc=g["gnuc"])
