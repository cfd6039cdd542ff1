"""Differential test of the LDS-staged streaming kernels (csrc/gemv_lds.hip: K5 ``gemv_rows_lds_kernel``, K8
``gemv_cols_lds_kernel``) against the fragment-shaped kernels they replace (csrc/gemv_mfma.hip), on shapes the
oracle would take minutes for (N up to 32, T up to 23, 12 ... 44 geometries per batch, both layouts): the same seeded
device data through two fresh processes, energies and forces compared.  (Both kernel families are held to the CPU
oracle on the shapes it can afford in test_gpu_bench_config.py / test_gpu_variants.py.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, name, env):
    e = dict(os.environ)
    e.update(env)
    out = str(tmp_path / (name + ".npz"))
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "lds_child.py"), out], cwd=REPO, env=e,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return np.load(out)


def test_lds_kernels_match_fragment_kernels(tmp_path):
    new = _run(tmp_path, "lds", {"EVC_ROWS_LDS": "1", "EVC_COLS_LDS": "1"})
    old = _run(tmp_path, "frag", {"EVC_ROWS_LDS": "0", "EVC_COLS_LDS": "0"})
    assert set(new.files) == set(old.files) and len(new.files) >= 16
    for k in new.files:
        a, b = new[k], old[k]
        assert a.shape == b.shape and np.isfinite(a).all(), k
        # energies to 1e-11 Ha, forces to 1e-10 Ha/Bohr: both families sum the same products in different orders
        tol = 1e-11 if k.startswith("E") else 1e-10
        assert np.abs(a - b).max() <= tol * max(1.0, float(np.abs(b).max())), (k, float(np.abs(a - b).max()))
