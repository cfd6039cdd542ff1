"""Device-backed mirror of ``evcont/ab_initio_eigenvector_continuation.py``."""
from __future__ import annotations

import os

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


def _eig_nonhermitian(H: np.ndarray, S: np.ndarray, layout: int):
    """The reference's non-Hermitian branch (:50-51,67-68,76-81) on the subspace matrix assembled by the
    device: pair layouts carry the two-body part in the lower triangle only, so the upper one is filled
    from it first; then ``scipy.linalg.eig(H, S)`` (a T x T problem, solved on the host exactly as the
    reference does) and the |Im| < 1e-5 filter."""
    import scipy.linalg
    H = np.array(H, dtype=np.float64)
    if layout in (5, 2):
        iu = np.triu_indices(H.shape[0])
        H[iu] = H.T[iu]
    vals, vecs = scipy.linalg.eig(H, np.asarray(S, dtype=np.float64))
    keep = np.abs(vals.imag) < 1.0e-5
    return vals[keep], vecs[:, keep]


def _select(vals, vecs, nroots, ground_state):
    if ground_state:
        k = int(np.argmin(vals.real))
        return float(vals[k].real), np.array(vecs[:, k].real)
    assert vals.shape[0] >= nroots                   # reference :166
    order = np.argsort(vals.real)[:nroots]
    return np.array(vals[order].real), np.array(vecs[:, order].real.T)


# How the resident copy of the training data is stored by the mol-level entry points (``*_OAO``,
# ``get_energy_with_grad``, ``MD_utils.get_scanner``):
#   "auto" (default)  8-fold compressed (``evaluator.DeviceTRDMs.compress_sym8_``: 3.7x fewer streamed bytes and the
#                     symmetric AO-side pipeline, the configuration ``bench.py`` measures) whenever that is exact AND
#                     indistinguishable from the reference for the caller: ``hermitian=True``, no predicted RDMs asked
#                     for, and AO integrals with the index symmetries of real two-electron integrals -- a PySCF ``Mole``
#                     has them by construction, an array-level molecule is checked at the first call for a training set
#                     (``integrals_have_symmetry``).  Everything else (``hermitian=False``,
#                     ``return_density_matrices=True``, ``predicted_two_rdm`` of a scanner, general tensors) runs on the
#                     layout the caller passed, from a second resident copy made on first use;
#   None / "none"     always the layout the caller passed;
#   "sym8"            always compressed (``predicted_two_rdm`` then is the 8-fold symmetrised 2-RDM; integrals without
#                     the symmetries raise ``EvcontHipError``).
def _mode_from_env():
    m = os.environ.get("EVCONT_AMD_COMPRESS", "auto").strip().lower()
    if m in ("", "none", "0", "off"):
        return None
    if m not in ("auto", "sym8"):
        raise ValueError(f"EVCONT_AMD_COMPRESS={m!r} (known: auto, sym8, none)")
    return m


_COMPRESS = _mode_from_env()
_auto_decisions = {}   # training-set key -> bool: integrals of its first geometry had the symmetries


def set_trdm_compression(mode) -> None:
    """``"auto"`` (default), ``"sym8"`` or ``None``; applies to training data uploaded from now on."""
    global _COMPRESS
    if mode not in (None, "sym8", "auto"):
        raise ValueError(f"unknown t-RDM compression {mode!r} (known: None, 'sym8', 'auto')")
    _COMPRESS = mode


def get_trdm_compression():
    return _COMPRESS


def integrals_have_symmetry(mol_or_ao, tol: float = 1.0e-9) -> bool:
    """Whether the AO integrals of a molecule have the index symmetries the compressed layout relies on (``eri``
    8-fold, ``eri_ip1[x,p,q,r,s] = eri_ip1[x,p,q,s,r]``).  A PySCF ``Mole`` (anything with ``intor``): yes, libcint
    integrals have them.  Array-level molecule: what it declares (``integral_symmetry``), else checked numerically (arrays handed over packed -- s4 / s2kl -- carry
    the symmetries of their packing; the pair exchange of a packed ``eri`` is still checked)."""
    declared = getattr(mol_or_ao, "integral_symmetry", None)
    if declared is not None:
        return bool(declared)
    if not is_array_mol(mol_or_ao):
        return True
    from .evaluator import check_integral_symmetry
    from ._lib import EvcontHipError
    n = int(np.asarray(mol_or_ao.S).shape[0])
    ip1 = getattr(mol_or_ao, "eri_ip1", None)
    if ip1 is not None and np.asarray(ip1).size == 0:
        ip1 = None
    try:
        check_integral_symmetry(np.asarray(mol_or_ao.eri), None if ip1 is None else np.asarray(ip1), n, tol)
    except EvcontHipError:
        return False
    return True


def resolve_compression(mode, one_RDM, two_RDM, S, mol, hermitian=True, want_rdms=False, n=None):
    """The storage (``None`` or ``"sym8"``) one call of a mol-level entry point uses under ``mode``
    (``"default"`` = ``get_trdm_compression()``)."""
    if mode == "default":
        mode = _COMPRESS
    if mode is None:
        return None
    if not hermitian:
        return None              # the eig branch works on the subspace matrix of the caller's layout
    if mode == "sym8":
        return "sym8"
    if want_rdms:
        return None              # the reference returns the un-symmetrised predicted 2-RDM
    key = cache.key_of(one_RDM, two_RDM, S, ("auto",))
    ok = _auto_decisions.get(key)
    if ok is None:
        ok = integrals_have_symmetry(mol)
        if len(_auto_decisions) > 64:
            _auto_decisions.clear()
        _auto_decisions[key] = ok
    return "sym8" if ok else None


def _trdms(one_RDM, two_RDM, S, compress=None) -> DeviceTRDMs:
    key = cache.key_of(one_RDM, two_RDM, S, ("trdms", compress))
    arrays = (one_RDM, two_RDM, S)
    t = cache.get(key, arrays)       # (block checksums of the host arrays are verified: cache.py)
    if t is None:
        t = cache.put(key, DeviceTRDMs(one_RDM, two_RDM, S, _dev(), compress=compress), arrays)
    return t


def _evaluator(one_RDM, two_RDM, S, natm: int, compress=None) -> ContinuationEvaluator:
    """``compress``: None or "sym8" (what ``resolve_compression`` returned for this call)."""
    assert compress in (None, "sym8")
    t = _trdms(one_RDM, two_RDM, S, compress)       # verified against the host arrays, or uploaded again
    key = ("evaluator", id(t), int(natm))
    ev = cache.get(key)
    if ev is None or ev.t is not t:
        ev = cache.put(key, ContinuationEvaluator(t, natm))
    return ev


def _solve_from_integrals(h1, h2, one_RDM, two_RDM, S, nroots, hermitian=True, ground_state=False):
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
    ev, vec, _, _, Hd = ops.subspace_solve(rows1, rows2.contiguous(), t.S, t.layout, nroots)
    if not hermitian:
        return _select(*_eig_nonhermitian(Hd.cpu().numpy(), t.S.cpu().numpy(), t.layout), nroots, ground_state)
    e = ev.cpu().numpy()
    if not np.all(np.isfinite(e)):
        raise np.linalg.LinAlgError("generalised eigenproblem failed (overlap not positive definite?)")
    return e, vec.cpu().numpy()


def approximate_ground_state(h1, h2, one_RDM, two_RDM, S, hermitian=True):
    """(E, c) of the lowest generalised eigenpair (reference :12-90)."""
    if not hermitian:
        return _solve_from_integrals(h1, h2, one_RDM, two_RDM, S, 1, hermitian=False, ground_state=True)
    e, c = _solve_from_integrals(h1, h2, one_RDM, two_RDM, S, 1)
    return float(e[0]), c[0].copy()


def approximate_multistate(h1, h2, one_RDM, two_RDM, S, nroots=1, hermitian=True):
    """(E[nroots], C[nroots,T]) lowest eigenpairs, rows S-orthonormal (reference :93-175)."""
    T = np.asarray(S).shape[0]
    assert T >= nroots                               # reference :166
    return _solve_from_integrals(h1, h2, one_RDM, two_RDM, S, int(nroots), hermitian=bool(hermitian))


def _oao(mol, one_RDM, two_RDM, S, nroots, hermitian=True, ground_state=False):
    ao = ao_arrays(mol, need_grad=False)
    # (the non-Hermitian branch works on the subspace matrix of the layout the caller passed)
    ev = _evaluator(one_RDM, two_RDM, S, int(np.asarray(ao.aoslices).shape[0]),
                    compress=resolve_compression("default", one_RDM, two_RDM, S, ao if is_array_mol(mol) else mol,
                                                 hermitian=hermitian))
    dao = DeviceAO.from_arrays(ao, ev.t.device, energy_only=True)
    res = ev.energies(dao, nroots)
    if hermitian:
        return res
    # same device pipeline (Loewdin, rotation, H build); only the T x T eigensolve differs
    e, c = _select(*_eig_nonhermitian(ev.hmat.cpu().numpy(), ev.t.S.cpu().numpy(), ev.t.layout), nroots, ground_state)
    return e + float(ao.enuc), c


def approximate_ground_state_OAO(mol, one_RDM, two_RDM, S, hermitian=True):
    """Total energy (incl. nuclear repulsion) and coefficients at the geometry of ``mol``
    (reference :178-211); Loewdin trafo, integral rotation, H build and eigensolve fused on the GPU."""
    if not hermitian:
        return _oao(mol, one_RDM, two_RDM, S, 1, hermitian=False, ground_state=True)
    e, c = _oao(mol, one_RDM, two_RDM, S, 1)
    return float(e[0]), c[0].copy()


def approximate_multistate_OAO(mol, one_RDM, two_RDM, S, nroots=1, hermitian=True):
    """Reference :214-250."""
    return _oao(mol, one_RDM, two_RDM, S, int(nroots), hermitian=bool(hermitian))
