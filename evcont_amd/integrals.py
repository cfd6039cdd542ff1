"""Adapter between the reference's ``mol`` argument and the array-level inputs of the
device path.

The reference pulls every AO quantity out of a PySCF ``Mole`` at the point of use
(``ab_initio_gradients_loewdin.py:25,147,283-284,336-339,369-370``).  Here the same queries
are made ONCE per geometry, on the host, and shipped to the device as a ``DeviceAO``.
PySCF is imported lazily and only when a real ``Mole`` is passed; anything exposing the
``AOArrays`` fields (``evcont_amd.synthetic.AOArrays``) is accepted as an "array-level mol",
which is how the tests and the benchmark drive the path without PySCF.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .synthetic import AOArrays

_FIELDS = ("S", "hcore", "eri")


def is_array_mol(mol) -> bool:
    return all(hasattr(mol, f) for f in _FIELDS) and not hasattr(mol, "intor")


def nao_of(mol) -> int:
    return int(mol.S.shape[0]) if is_array_mol(mol) else int(mol.nao)


def ao_arrays(mol, need_grad: bool = True) -> AOArrays:
    """AO arrays of one geometry (host, float64)."""
    if is_array_mol(mol):
        return mol
    from pyscf import scf, grad  # lazy: only for real molecules
    n = int(mol.nao)
    S = np.asarray(mol.intor("int1e_ovlp"), dtype=np.float64)
    hcore = np.asarray(scf.hf.get_hcore(mol), dtype=np.float64)
    eri = np.asarray(mol.intor("int2e"), dtype=np.float64).reshape(n, n, n, n)
    sl = np.asarray([[s[2], s[3]] for s in mol.aoslice_by_atom()], dtype=np.int64)
    enuc = float(mol.energy_nuc())
    if not need_grad:
        z = np.zeros
        return AOArrays(S, hcore, eri, z((3, n, n)), z((len(sl), 3, n, n)), z((3, 0)), sl, enuc, z((len(sl), 3)))
    g = grad.RHF(scf.RHF(mol))
    gen = g.hcore_generator()
    dh = np.asarray([gen(i) for i in range(mol.natm)], dtype=np.float64)          # (A,3,N,N)
    ipovlp = np.asarray(mol.intor("int1e_ipovlp", comp=3), dtype=np.float64)
    ip1 = np.asarray(mol.intor("int2e_ip1", comp=3), dtype=np.float64).reshape(3, n, n, n, n)
    gnuc = np.asarray(g.grad_nuc(), dtype=np.float64)
    return AOArrays(S, hcore, eri, ipovlp, dh, ip1, sl, enuc, gnuc)


def aoslices_of(mol) -> np.ndarray:
    """(A,2) [start, stop) of every atom's AOs."""
    if is_array_mol(mol):
        return np.asarray(mol.aoslices, dtype=np.int64).reshape(-1, 2)
    return np.asarray([[s[2], s[3]] for s in mol.aoslice_by_atom()], dtype=np.int64)


def stage_mol(mol, hev) -> None:
    """Fill the pinned staging buffers of a ``hosted.HostedEvaluator`` with the AO integrals of ``mol``.  A PySCF
    ``Mole`` writes the two large arrays straight into them (``intor(..., out=)``), packed as the device side wants
    them (``aosym="s4"`` / ``"s2kl"``) -- no full N^4 array and no staging copy on the host."""
    if is_array_mol(mol):
        hev.stage(mol)
        return
    from pyscf import scf, grad
    st = hev.staging()
    n = int(mol.nao)
    np.copyto(st["S"], mol.intor("int1e_ovlp"))
    np.copyto(st["hcore"], scf.hf.get_hcore(mol))
    np.copyto(st["ipovlp"], mol.intor("int1e_ipovlp", comp=3))
    st["enuc"][0] = float(mol.energy_nuc())
    g = grad.RHF(scf.RHF(mol))
    gen = g.hcore_generator()
    for ia in range(mol.natm):
        np.copyto(st["dhcore"][ia], gen(ia))
    np.copyto(st["gnuc"], g.grad_nuc())

    def fill(dst, name, **kw):
        try:
            out = mol.intor(name, out=dst, **kw)
        except TypeError:                 # an intor without `out=` support: one extra host copy
            out = mol.intor(name, **kw)
        if out is not dst and not np.shares_memory(out, dst):
            np.copyto(dst, np.asarray(out).reshape(dst.shape))

    if hev.packed:
        fill(st["eri"], "int2e", aosym="s4")
        fill(st["eri_ip1"], "int2e_ip1", comp=3, aosym="s2kl")
    else:
        fill(st["eri"].reshape(n * n, n * n), "int2e")
        fill(st["eri_ip1"].reshape(3, n * n, n * n), "int2e_ip1", comp=3)


def energy_nuc(mol) -> float:
    return float(mol.enuc) if is_array_mol(mol) else float(mol.energy_nuc())


def grad_nuc(mol) -> np.ndarray:
    if is_array_mol(mol):
        return np.asarray(mol.gnuc, dtype=np.float64)
    from pyscf import scf, grad
    return np.asarray(grad.RHF(scf.RHF(mol)).grad_nuc(), dtype=np.float64)
