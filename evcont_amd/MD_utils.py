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
                                                 get_trdm_compression, integrals_have_symmetry)
from .electron_integral_utils import get_basis, get_integrals  # noqa: F401 (re-export)
from .evaluator import ContinuationEvaluator, DeviceAO
from .hosted import HostedEvaluator
from .integrals import ao_arrays, aoslices_of, energy_nuc, grad_nuc, stage_mol


def _grad_scanner_base():
    try:
        from pyscf import lib
        return lib.GradScanner
    except Exception:  # PySCF absent: the scanner is still a plain callable (tests, array-level mols)
        return object


def get_scanner(mol, one_rdm, two_rdm, overlap, hermitian=True, compress="default", device_trdms=None):
    """Fake PySCF gradient scanner driven by the continuation (reference :20-57): ``scanner(mol)``
    returns ``(E_tot, grad)`` and stores the predicted RDMs on ``scanner.base``.

    ``compress``: storage of the resident training data, ``None``, ``"sym8"``, ``"auto"`` or ``"default"``
    (= ``set_trdm_compression``, "auto" unless changed).  "auto": the 8-fold compressed copy and the packed s4 / s2kl
    integrals for the per-step ``(E, grad)`` when ``hermitian`` and the first molecule's integrals have the symmetries
    of real ones (a PySCF ``Mole`` always; array-level molecules are checked once), with ``predicted_two_rdm``
    still the reference's un-symmetrised 2-RDM (evaluated on the caller's layout when it is read); with ``"sym8"``
    the stored predicted 2-RDM is the 8-fold symmetrised one.
    ``device_trdms``: training data already resident on the device (``trdm_io.load_pair_directories`` /
    ``load_checkpoint``, a container's ``device_trdms()``): used instead of uploading the host arrays."""
    if compress == "default":
        compress = get_trdm_compression()
    if compress not in (None, "sym8", "auto"):
        raise ValueError(f"unknown t-RDM compression {compress!r} (known: None, 'sym8', 'auto', 'default')")
    auto = compress == "auto"

    have_data = device_trdms is not None or (one_rdm is not None and two_rdm is not None and overlap is not None)

    class Base:
        """``scanner.base`` of the reference (:26-32).  The predicted RDMs are fetched from the device when they are
        read: integrator callbacks use ``predicted_one_rdm`` (``04_Zundel_continuation_MD.py:145``); nothing in the
        reference reads the N^4 ``predicted_two_rdm`` during a run, so it is produced on first access (a second
        evaluation of the current geometry with the unpacked 2-RDM as an output) instead of being shipped every step."""
        converged = True
        ovlp = overlap
        one_trdm = one_rdm
        two_trdm = two_rdm

        def __init__(self, owner):
            self._owner = owner

        @property
        def predicted_one_rdm(self):
            return self._owner._one()

        @property
        def predicted_two_rdm(self):
            return self._owner._two()

    class Scanner(_grad_scanner_base()):
        def __init__(self):
            self.mol = mol
            self.base = Base(self)
            self._compress = None if auto else compress   # "auto": decided at the first call (needs integrals)
            self._decided = not auto
            self._hev = None          # HostedEvaluator (Hermitian path)
            self._full = None         # ContinuationEvaluator with every output (lazy 2-RDM, hermitian=False)
            self._last = None         # (mol, D, G) of the last call as far as known

        # -- lazily fetched outputs -----------------------------------------------------------------
        def _one(self):
            if self._last is None:
                return None
            if self._last[1] is None and self._hev is not None:
                self._last[1] = self._hev.predicted_one_rdm()
            return self._last[1]

        def _two(self):
            if self._last is None:
                return None
            if self._last[2] is None:
                self._last[2] = self._full_eval(self._last[0])[3]
            return self._last[2]

        def _full_eval(self, m):
            ao = ao_arrays(m, need_grad=True)
            if self._full is None:
                # "auto": the predicted 2-RDM as the reference returns it -- from the layout the caller passed
                t = device_trdms if device_trdms is not None else \
                    _trdms(one_rdm, two_rdm, overlap, None if auto else self._compress)
                self._full = ContinuationEvaluator(t, int(np.asarray(ao.aoslices).shape[0]))
            return self._full.energy_with_grad(DeviceAO.from_arrays(ao, self._full.t.device), True)

        def __call__(self, mol):
            self.mol = mol
            if not have_data:
                return energy_nuc(mol), grad_nuc(mol)
            if not hermitian:
                if device_trdms is not None and (one_rdm is None or two_rdm is None or overlap is None):
                    # training data resident on the device only: the eig branch on ITS layout (not the compressed one)
                    if self._full is None:
                        self._full = ContinuationEvaluator(device_trdms, len(aoslices_of(mol)))
                    ao = ao_arrays(mol, need_grad=True)
                    en, grad, rdm_o, rdm_t = self._full.energy_with_grad_nonhermitian(
                        DeviceAO.from_arrays(ao, device_trdms.device), True)
                else:
                    en, grad, rdm_o, rdm_t = get_energy_with_grad(
                        mol, one_rdm, two_rdm, overlap, hermitian=hermitian, return_density_matrices=True)
                self._last = [mol, rdm_o, rdm_t]
                return en, grad
            # called once per MD step on slowly moving geometries: the scanner owns a hosted evaluator (pinned
            # staging, eager two-stream enqueue of uploads + kernels + download -- replaying the step as one HIP graph
            # is an opt-in, EVCONT_AMD_HOSTED_GRAPH=1, measured slower --, eigensolvers warm-started from the previous
            # step; with the compressed layout the two large integral arrays are requested / staged packed: 12 instead
            # of 26.6 MB at H30)
            if self._hev is None:
                if not self._decided:
                    self._compress = "sym8" if integrals_have_symmetry(mol) else None
                    self._decided = True
                t = device_trdms if device_trdms is not None else _trdms(one_rdm, two_rdm, overlap, self._compress)
                sl = aoslices_of(mol)
                self._hev = HostedEvaluator(t, len(sl), sl, warm_start=True)
            stage_mol(mol, self._hev)
            en, grad = self._hev.run()
            self._last = [mol, None, None]
            return en, grad

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


KB_HARTREE = 3.166811563e-6   # Boltzmann constant in Hartree / K


def nve_velocity_verlet(scanner, init_mol, dt=10.0, steps=10, veloc=None, trajectory_output=None,
                        energy_output=None, callback=None, thermostat=None):
    """Velocity-Verlet NVE propagation of an array-level molecule with ``scanner(mol) -> (E, grad)``.
    Returns one frame per step: ``{"coord", "veloc", "epot", "ekin", "time"}`` (frame 0 = initial geometry);
    the force of a step's end point is reused as the next step's start, so there is exactly one
    energy+force evaluation per step.

    ``callback(locals())`` is called after every frame with ``mol`` and ``scanner`` among the keys, as PySCF's
    integrators do (``04_Zundel_continuation_MD.py:140-177`` reads ``locals["scanner"].base.predicted_one_rdm``);
    ``thermostat=(T_kelvin, taut)`` rescales the velocities after every step like ``md.integrators.NVTBerendson``."""
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
        if callable(callback):
            frame, iteration = frames[-1], k   # noqa: F841 (exposed to the callback through locals())
            callback(locals())
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
        if thermostat is not None:
            T_target, taut = thermostat
            T_now = float(np.sum(m * v * v)) / (3 * R.shape[0] * KB_HARTREE)
            if T_now > 0.0:
                v = v * np.sqrt(1.0 + (T_target / T_now - 1.0) * dt / taut)
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
