"""Device-backed mirror of the evaluator-facing part of ``evcont/MD_utils.py``.

``get_scanner`` (reference :20-57) is the harness PySCF's MD integrators call once per step;
``get_trajectory`` (:60-125) is a thin wrapper around ``pyscf.md.NVE`` (or the velocity-Verlet
driver below for array-level molecules).  The active-learning driver ``converge_EVCont_MD``
(:128-502) lives in ``evcont_amd/active_learning.py``.
"""
from __future__ import annotations

import numpy as np

from .ab_initio_gradients_loewdin import get_energy_with_grad
from .ab_initio_eigenvector_continuation import (approximate_ground_state_OAO, _trdms,  # noqa: F401 (re-export)
                                                 get_trdm_compression)
from .electron_integral_utils import get_basis, get_integrals  # noqa: F401 (re-export)
from .evaluator import ContinuationEvaluator, DeviceAO
from .integrals import ao_arrays, energy_nuc, grad_nuc


def _grad_scanner_base():
    try:
        from pyscf import lib
        return lib.GradScanner
    except Exception:  # PySCF absent: the scanner is still a plain callable (tests, array-level mols)
        return object


def get_scanner(mol, one_rdm, two_rdm, overlap, hermitian=True, compress="default"):
    """Fake PySCF gradient scanner driven by the continuation (reference :20-57): ``scanner(mol)``
    returns ``(E_tot, grad)`` and stores the predicted RDMs on ``scanner.base``.

    ``compress``: storage of the resident training data, ``None``, ``"sym8"`` or ``"default"``
    (= ``set_trdm_compression``); with ``"sym8"`` the stored predicted 2-RDM is the 8-fold symmetrised one."""
    if compress == "default":
        compress = get_trdm_compression()

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
            self._ev = None

        def __call__(self, mol):
            self.mol = mol
            if one_rdm is not None and two_rdm is not None and overlap is not None:
                if not hermitian:
                    en, grad, rdm_o, rdm_t = get_energy_with_grad(
                        mol, one_rdm, two_rdm, overlap, hermitian=hermitian, return_density_matrices=True)
                else:
                    # the scanner is called once per MD step on slowly moving geometries: it owns an evaluator
                    # whose eigensolvers start from the previous step's eigenvectors (EVC_FLAG_WARM_START)
                    ao = ao_arrays(mol, need_grad=True)
                    if self._ev is None:
                        self._ev = ContinuationEvaluator(_trdms(one_rdm, two_rdm, overlap, compress),
                                                         int(np.asarray(ao.aoslices).shape[0]), warm_start=True)
                    # (compressed layout: int2e_ip1 travels packed in its last two AO indices, half the upload)
                    pack = self._ev.t.layout == 8 and self._ev.t.n <= 32
                    en, grad, rdm_o, rdm_t = self._ev.energy_with_grad(
                        DeviceAO.from_arrays(ao, self._ev.t.device, pack_ip1=pack), return_density_matrices=True)
                self.base.predicted_one_rdm = rdm_o
                self.base.predicted_two_rdm = rdm_t
                return en, grad
            return energy_nuc(mol), grad_nuc(mol)

    return Scanner()


def get_trajectory(init_mol, overlap, one_rdm, two_rdm, dt=10.0, steps=10, init_veloc=None, hermitian=True,
                   trajectory_output=None, energy_output=None, compress="default"):
    """NVE trajectory from the continuation (reference :60-125).  Single process: the reference's
    rank-0-computes / Bcast split exists only to coexist with MPI-parallel training code.

    PySCF molecules are propagated by ``pyscf.md.NVE`` exactly as in the reference; array-level molecules
    that can be rebuilt at new coordinates (``with_coords``, e.g. ``evcont_amd.hchain.HChainMol``) by the
    velocity-Verlet integrator below, with the same conventions (Bohr, atomic time units, frame 0 = the
    initial geometry, ``steps`` frames)."""
    scanner_fun = get_scanner(init_mol, one_rdm, two_rdm, overlap, hermitian=hermitian, compress=compress)
    if hasattr(init_mol, "with_coords"):
        frames = nve_velocity_verlet(scanner_fun, init_mol, dt=dt, steps=steps, veloc=init_veloc,
                                     trajectory_output=trajectory_output, energy_output=energy_output)
        return np.array([f["coord"] for f in frames])
    from pyscf import md  # needs PySCF's integrator
    frames = []
    integ = md.NVE(scanner_fun, dt=dt, steps=steps, veloc=init_veloc, incore_anyway=True, frames=frames,
                   trajectory_output=trajectory_output, energy_output=energy_output, verbose=0)
    integ.run()
    return np.array([frame.coord for frame in frames])


AMU2AU = 1822.888486209      # atomic mass unit in electron masses (CODATA 2018)


def nve_velocity_verlet(scanner, init_mol, dt=10.0, steps=10, veloc=None, trajectory_output=None,
                        energy_output=None):
    """Velocity-Verlet NVE propagation of an array-level molecule with ``scanner(mol) -> (E, grad)``.
    Returns one frame per step: ``{"coord", "veloc", "epot", "ekin", "time"}`` (frame 0 = initial geometry);
    the force of a step's end point is reused as the next step's start, so there is exactly one
    energy+force evaluation per step."""
    R = np.array(init_mol.atom_coords(), dtype=np.float64)
    m = (np.asarray(init_mol.atom_mass_list(), dtype=np.float64) * AMU2AU)[:, None]
    v = np.zeros_like(R) if veloc is None else np.array(veloc, dtype=np.float64)
    mol = init_mol
    e, g = scanner(mol)
    frames = []
    fe = open(energy_output, "w") if isinstance(energy_output, str) else energy_output
    ft = open(trajectory_output, "w") if isinstance(trajectory_output, str) else trajectory_output
    for k in range(steps):
        ekin = 0.5 * float(np.sum(m * v * v))
        frames.append({"coord": R.copy(), "veloc": v.copy(), "epot": float(e), "ekin": ekin, "time": k * dt})
        if fe is not None:
            fe.write(f"{k * dt:14.6f} {e:18.10f} {ekin:18.10f} {e + ekin:18.10f}\n")
        if ft is not None:
            ft.write(f"{R.shape[0]}\nMD time {k * dt:.6f} a.u., coordinates in Bohr\n")
            for xyz in R:
                ft.write("H %18.10f %18.10f %18.10f\n" % tuple(xyz))
        if k == steps - 1:
            break
        a = -np.asarray(g) / m
        R = R + dt * v + 0.5 * dt * dt * a
        mol = init_mol.with_coords(R)
        e, g = scanner(mol)
        v = v + 0.5 * dt * (a - np.asarray(g) / m)
    for f, given in ((fe, energy_output), (ft, trajectory_output)):
        if f is not None and isinstance(given, str):
            f.close()
    return frames


def converge_EVCont_MD(EVCont_obj, init_mol, steps=100, dt=1, convergence_thresh=1.0e-3,
                       prune_irrelevant_data=False, trn_times=[], data_addition="farthest_point_ham", **kwargs):
    """Active-learning driver (reference :128-502): see ``evcont_amd.active_learning.converge_EVCont_MD``,
    which evaluates the loop's re-evaluation and selection steps as batched device calls."""
    from .active_learning import converge_EVCont_MD as _impl
    return _impl(EVCont_obj, init_mol, steps=steps, dt=dt, convergence_thresh=convergence_thresh,
                 prune_irrelevant_data=prune_irrelevant_data, trn_times=list(trn_times),
                 data_addition=data_addition, **kwargs)
