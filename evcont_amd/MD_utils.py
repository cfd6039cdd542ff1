"""Device-backed mirror of the evaluator-facing part of ``evcont/MD_utils.py``.

``get_scanner`` (reference :20-57) is the harness PySCF's MD integrators call once per step;
``get_trajectory`` (:60-125) is a thin wrapper around ``pyscf.md.NVE``.  The active-learning
driver ``converge_EVCont_MD`` (:128-502) is host-side orchestration with file checkpoints and
MPI broadcasts and is outside this build's scope; scripts that use it keep
importing it from the reference and get the accelerated evaluator through the functions here.
"""
from __future__ import annotations

import numpy as np

from .ab_initio_gradients_loewdin import get_energy_with_grad
from .ab_initio_eigenvector_continuation import approximate_ground_state_OAO  # noqa: F401 (re-export)
from .electron_integral_utils import get_basis, get_integrals  # noqa: F401 (re-export)
from .integrals import energy_nuc, grad_nuc


def _grad_scanner_base():
    try:
        from pyscf import lib
        return lib.GradScanner
    except Exception:  # PySCF absent: the scanner is still a plain callable (tests, array-level mols)
        return object


def get_scanner(mol, one_rdm, two_rdm, overlap, hermitian=True):
    """Fake PySCF gradient scanner driven by the continuation (reference :20-57): ``scanner(mol)``
    returns ``(E_tot, grad)`` and stores the predicted RDMs on ``scanner.base``."""

    class Base:
        converged = True
        ovlp = overlap
        one_trdm = one_rdm
        two_trdm = two_rdm
        predicted_one_rdm = None
        predicted_two_rdm = None

    class Scanner(_grad_scanner_base()):
        def __init__(self):
            self.mol = mol
            self.base = Base()

        def __call__(self, mol):
            self.mol = mol
            if one_rdm is not None and two_rdm is not None and overlap is not None:
                en, grad, rdm_o, rdm_t = get_energy_with_grad(
                    mol, one_rdm, two_rdm, overlap, hermitian=hermitian, return_density_matrices=True)
                self.base.predicted_one_rdm = rdm_o
                self.base.predicted_two_rdm = rdm_t
                return en, grad
            return energy_nuc(mol), grad_nuc(mol)

    return Scanner()


def get_trajectory(init_mol, overlap, one_rdm, two_rdm, dt=10.0, steps=10, init_veloc=None, hermitian=True,
                   trajectory_output=None, energy_output=None):
    """NVE trajectory from the continuation (reference :60-125).  Single process: the reference's
    rank-0-computes / Bcast split exists only to coexist with MPI-parallel training code."""
    from pyscf import md  # needs PySCF's integrator
    scanner_fun = get_scanner(init_mol, one_rdm, two_rdm, overlap, hermitian=hermitian)
    frames = []
    integ = md.NVE(scanner_fun, dt=dt, steps=steps, veloc=init_veloc, incore_anyway=True, frames=frames,
                   trajectory_output=trajectory_output, energy_output=energy_output, verbose=0)
    integ.run()
    return np.array([frame.coord for frame in frames])


def converge_EVCont_MD(*args, **kwargs):
    raise NotImplementedError(
        "converge_EVCont_MD (reference MD_utils.py:128-502) is host orchestration outside this build's scope; "
        "use the reference's driver with evcont_amd's evaluator functions")
