"""Device-backed mirror of ``evcont/ab_initio_eigenvector_continuation.py``."""
from __future__ import annotations

import numpy as np
import torch

from . import cache, ops
from .electron_integral_utils import (  # re-exported, scripts import them from here (04_Zundel...py:13-15)
    get_basis,
    get_integrals,
    compress_electron_exchange_symmetry,
)
from .evaluator import DeviceTRDMs, DeviceAO, ContinuationEvaluator, _dev
from .integrals import ao_arrays, energy_nuc, is_array_mol


def _nonhermitian_unsupported():
    raise NotImplementedError(
        "hermitian=False (scipy.linalg.eig branch, reference :76-78) is not implemented on the device; "
        "every script of the reference uses hermitian=True")


def _trdms(one_RDM, two_RDM, S) -> DeviceTRDMs:
    key = cache.key_of(one_RDM, two_RDM, S, ("trdms",))
    t = cache.get(key)
    if t is None:
        t = cache.put(key, DeviceTRDMs(one_RDM, two_RDM, S, _dev()))
    return t


def _evaluator(one_RDM, two_RDM, S, natm: int) -> ContinuationEvaluator:
    key = cache.key_of(one_RDM, two_RDM, S, ("evaluator", int(natm)))
    ev = cache.get(key)
    if ev is None:
        ev = cache.put(key, ContinuationEvaluator(_trdms(one_RDM, two_RDM, S), natm))
    return ev


def _solve_from_integrals(h1, h2, one_RDM, two_RDM, S, nroots):
    """H_ab from given OAO integrals (reference :38-68) + eigh(H, S) (:73-88), all on the device."""
    two_RDM = np.asarray(two_RDM) if not torch.is_tensor(two_RDM) else two_RDM
    assert two_RDM.ndim in (6, 5, 3, 2)          # reference: `assert False` otherwise (:70-71)
    t = _trdms(one_RDM, two_RDM, S)
    d = t.device
    h2d = ops.to_device(np.asarray(h2, dtype=np.float64).reshape((t.n,) * 4), d)
    h1d = ops.to_device(np.asarray(h1, dtype=np.float64), d).reshape(-1)
    if t.layout in (3, 2):
        v, alpha = ops.pack_pair_sym(h2d, 0.5, pad_to=t.ld), 1.0
    else:
        v, alpha = h2d.reshape(-1), 0.5
    rows2 = ops.gemv_rows(t.two, t.cols, v, alpha)[: t.rows_total]
    rows1 = ops.gemv_rows(t.one, t.n * t.n, h1d)          # t.one rows are padded to an even length
    ev, vec, _, _, _ = ops.subspace_solve(rows1, rows2.contiguous(), t.S, t.layout, nroots)
    e = ev.cpu().numpy()
    if not np.all(np.isfinite(e)):
        raise np.linalg.LinAlgError("generalised eigenproblem failed (overlap not positive definite?)")
    return e, vec.cpu().numpy()


def approximate_ground_state(h1, h2, one_RDM, two_RDM, S, hermitian=True):
    """(E, c) of the lowest generalised eigenpair (reference :12-90)."""
    if not hermitian:
        _nonhermitian_unsupported()
    e, c = _solve_from_integrals(h1, h2, one_RDM, two_RDM, S, 1)
    return float(e[0]), c[0].copy()


def approximate_multistate(h1, h2, one_RDM, two_RDM, S, nroots=1, hermitian=True):
    """(E[nroots], C[nroots,T]) lowest eigenpairs, rows S-orthonormal (reference :93-175)."""
    if not hermitian:
        _nonhermitian_unsupported()
    T = np.asarray(S).shape[0]
    assert T >= nroots                               # reference :166
    return _solve_from_integrals(h1, h2, one_RDM, two_RDM, S, int(nroots))


def _oao(mol, one_RDM, two_RDM, S, nroots):
    ao = ao_arrays(mol, need_grad=False)
    ev = _evaluator(one_RDM, two_RDM, S, int(np.asarray(ao.aoslices).shape[0]))
    dao = DeviceAO.from_arrays(ao, ev.t.device, energy_only=True)
    return ev.energies(dao, nroots)


def approximate_ground_state_OAO(mol, one_RDM, two_RDM, S, hermitian=True):
    """Total energy (incl. nuclear repulsion) and coefficients at the geometry of ``mol``
    (reference :178-211); Loewdin trafo, integral rotation, H build and eigensolve fused on the GPU."""
    if not hermitian:
        _nonhermitian_unsupported()
    e, c = _oao(mol, one_RDM, two_RDM, S, 1)
    return float(e[0]), c[0].copy()


def approximate_multistate_OAO(mol, one_RDM, two_RDM, S, nroots=1, hermitian=True):
    """Reference :214-250."""
    if not hermitian:
        _nonhermitian_unsupported()
    return _oao(mol, one_RDM, two_RDM, S, int(nroots))
