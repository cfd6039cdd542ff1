"""Device-backed mirror of ``evcont/electron_integral_utils.py`` (same names, arguments and
return conventions; numpy in, numpy out).  ``mol`` may be a PySCF ``Mole`` or an array-level
stand-in (``evcont_amd.synthetic.AOArrays``)."""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .evaluator import _dev
from .integrals import ao_arrays, is_array_mol


def _up(x):
    return ops.to_device(np.asarray(x, dtype=np.float64), _dev())


def get_loewdin_trafo(overlap_mat):
    """X = S^(-1/2) with the 1e-15 eigenvalue guard (reference :6-18); Jacobi eigensolver on the GPU."""
    X, _, _ = ops.loewdin(_up(overlap_mat))
    return X.cpu().numpy()


def transform_integrals(h1, h2, trafo):
    """h1'[a,b] = sum_ij h1[i,j] T[a,i] T[b,j];  h2'[a,b,c,d] = sum h2[i,j,k,l] T[a,i] T[b,j] T[c,k] T[d,l]
    (the evident intent of reference :21-35, whose einsum calls are malformed and raise ValueError)."""
    T = np.asarray(trafo, dtype=np.float64)
    h1n = np.asarray(h1, dtype=np.float64)
    h2n = np.asarray(h2, dtype=np.float64)
    if h1n.ndim != 2 or h2n.ndim != 4:
        raise ValueError("transform_integrals: only un-batched h1 (N,N) and h2 (N,N,N,N) are supported")
    h1o = T @ h1n @ T.T
    h2o = ops.four_index_transform(_up(h2n), _up(T), transposed=True).cpu().numpy()
    return h1o, h2o


def compress_electron_exchange_symmetry(h2, diag_multiplier=1.0):
    """(N,N,N,N) -> packed lower triangle of the (N^2,N^2) matrix (reference :38-66); out of place."""
    h2 = np.asarray(h2)
    assert np.all(np.array(h2.shape) == h2.shape[0])
    return ops.pack_pair_sym(_up(h2), float(diag_multiplier)).cpu().numpy()


def restore_electron_exchange_symmetry(h2, norb):
    """Inverse of the packing with multiplier 1 (reference :69-88)."""
    return ops.unpack_pair_sym(_up(h2), int(norb)).cpu().numpy()


def get_basis(mol, basis_type="OAO"):
    """AO->MO coefficients of an orthogonal basis (reference :91-119).  Only the OAO branch is on
    the accelerated path; the canonical / split-localised branches are PySCF host code."""
    if basis_type == "OAO":
        S = mol.S if is_array_mol(mol) else mol.intor("int1e_ovlp")
        return get_loewdin_trafo(S)
    from pyscf import scf, lo  # host-side PySCF, exactly as the reference does
    myhf = scf.RHF(mol)
    _ = myhf.scf()
    basis = myhf.mo_coeff
    if basis_type == "split":
        loc = lo.Boys(mol, basis[:, : mol.nelec[0]])
        loc.init_guess = None
        occ = loc.kernel()
        loc = lo.Boys(mol, basis[:, mol.nelec[0]:])
        loc.init_guess = None
        vrt = loc.kernel()
        basis = np.concatenate((occ, vrt), axis=1)
    else:
        assert basis_type == "canonical"
    return basis


def get_integrals(mol, basis):
    """h1 = C^T hcore C, h2 = four-index rotation of the AO ERI, s1 (N,N,N,N) (reference :122-138;
    the reference delegates the 4-index step to pyscf.ao2mo, here FP64 MFMA on the GPU)."""
    ao = ao_arrays(mol, need_grad=False)
    Cm = np.asarray(basis, dtype=np.float64)
    h1 = np.linalg.multi_dot((Cm.T, ao.hcore, Cm))
    if Cm.shape[0] != Cm.shape[1]:
        raise NotImplementedError("get_integrals: rectangular bases are not supported by the device transform")
    h2 = ops.four_index_transform(_up(ao.eri), _up(Cm), transposed=False).cpu().numpy()
    return h1, h2
