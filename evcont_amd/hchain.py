"""Self-contained AO integrals (and their first nuclear derivatives) for molecules built from
s-type contracted Gaussians — hydrogen chains/clusters in STO-3G, the systems of BASELINE
configs 0-2 (``scripts/PES_H_chain``, ``scripts/MD/H30``).

This replaces, for those systems, the PySCF/libcint calls the hot path consumes as *inputs*
(``ab_initio_gradients_loewdin.py:25,147,283-284,336-339,369-370``):

    S        int1e_ovlp                 (N,N)
    hcore    scf.hf.get_hcore           (N,N)        kinetic + nuclear attraction
    eri      int2e                      (N,N,N,N)    chemists' (pq|rs)
    ipovlp   int1e_ipovlp               (3,N,N)      <d/dr mu | nu>   (electron-coordinate gradient)
    dhcore   grad.RHF.hcore_generator   (A,3,N,N)    d hcore / d R_A  (basis functions AND nucleus A move)
    eri_ip1  int2e_ip1                  (3,N,N,N,N)  (d/dr mu nu | kappa lambda)
    enuc, gnuc                                        nuclear repulsion and its gradient

Closed forms for s primitives (Boys F0/F1).  For an s function a gradient with respect to the
electron coordinate is minus the gradient with respect to its centre, which is what fixes the
signs of ``ipovlp`` / ``eri_ip1`` relative to the nuclear derivatives; every derivative here is
checked against central finite differences in ``tests/test_hchain_physics.py``.

Host code (numpy): these are inputs of the accelerated path, produced once per geometry, exactly
where the reference calls libcint.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np
from scipy.special import erf

from .synthetic import AOArrays

# STO-3G hydrogen 1s (zeta = 1.24): exponents and contraction coefficients of normalised primitives
STO3G_H_EXPONENTS = (3.42525091, 0.62391373, 0.16885540)
STO3G_H_COEFFICIENTS = (0.15432897, 0.53532814, 0.44463454)
# STO-6G hydrogen 1s (the basis of scripts/PES_H_chain)
STO6G_H_EXPONENTS = (35.52322122, 6.513143725, 1.822142904, 0.6259552659, 0.2430767471, 0.1001124280)
STO6G_H_COEFFICIENTS = (0.009163596281, 0.04936149294, 0.1685383049, 0.3705627997, 0.4164915298, 0.1303340841)


def boys01(t: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """F0(t) and F1(t) = -F0'(t)."""
    t = np.asarray(t, dtype=np.float64)
    small = t < 1e-2
    ts = np.where(small, t, 1.0)
    # Taylor: F_n(t) = sum_k (-t)^k / (k! (2n+2k+1))
    f0s = np.zeros_like(ts)
    f1s = np.zeros_like(ts)
    term = np.ones_like(ts)
    for k in range(9):
        f0s += term / (2 * k + 1)
        f1s += term / (2 * k + 3)
        term = term * (-ts) / (k + 1)
    tl = np.where(small, 1.0, t)
    rt = np.sqrt(tl)
    f0l = 0.5 * np.sqrt(np.pi) / rt * erf(rt)
    f1l = (f0l - np.exp(-tl)) / (2.0 * tl)
    return np.where(small, f0s, f0l), np.where(small, f1s, f1l)


@dataclass
class HChainMol(AOArrays):
    """``AOArrays`` of a geometry plus what the training-state generators ask a ``mol`` for."""
    coords: Optional[np.ndarray] = None   # (A,3) Bohr
    nelec: Tuple[int, int] = (0, 0)
    charges: Optional[np.ndarray] = None
    exponents: Tuple[float, ...] = STO3G_H_EXPONENTS
    coefficients: Tuple[float, ...] = STO3G_H_COEFFICIENTS

    def energy_nuc(self) -> float:
        return float(self.enuc)

    def atom_coords(self) -> np.ndarray:
        return np.array(self.coords, copy=True)

    def atom_mass_list(self) -> np.ndarray:
        """Atomic masses in amu (hydrogen-like centres: 1.008 per unit charge is only right for H)."""
        return np.full(self.natm, 1.008)

    def with_coords(self, coords, need_grad: bool = True) -> "HChainMol":
        """The same molecule (basis, charges, electrons) at new nuclear positions (Bohr)."""
        return s_gaussian_mol(coords, self.charges, self.exponents, self.coefficients, need_grad, self.nelec)


def s_gaussian_mol(coords, charges: Optional[Sequence[float]] = None,
                   exponents: Sequence[float] = STO3G_H_EXPONENTS,
                   coefficients: Sequence[float] = STO3G_H_COEFFICIENTS,
                   need_grad: bool = True, nelec: Optional[Tuple[int, int]] = None) -> HChainMol:
    """AO arrays for one contracted s function on every atom (``coords`` (A,3) in Bohr, ``charges`` default 1)."""
    R = np.ascontiguousarray(np.asarray(coords, dtype=np.float64).reshape(-1, 3))
    A = R.shape[0]
    Z = np.ones(A) if charges is None else np.asarray(charges, dtype=np.float64)
    ex = np.asarray(exponents, dtype=np.float64)
    co = np.asarray(coefficients, dtype=np.float64)
    K = len(ex)
    n, Np = A, A * K
    a = np.tile(ex, A)                                      # (Np,) primitive exponents
    owner = np.repeat(np.arange(A), K)                      # contracted function (= atom) of each primitive
    Rp = R[owner]                                           # (Np,3)
    cn = np.tile(co, A) * (2.0 * a / np.pi) ** 0.75         # coefficient x primitive norm
    Cm = np.zeros((Np, n))
    Cm[np.arange(Np), owner] = cn

    # ---- primitive pair quantities
    pp = a[:, None] + a[None, :]
    mu = a[:, None] * a[None, :] / pp
    AB = Rp[:, None, :] - Rp[None, :, :]                    # (Np,Np,3)  A - B
    R2 = np.sum(AB * AB, axis=-1)
    Kab = np.exp(-mu * R2)
    P = (a[:, None, None] * Rp[:, None, :] + a[None, :, None] * Rp[None, :, :]) / pp[:, :, None]

    def contract2(M):                                        # (..., Np, Np) -> (..., n, n)
        return np.einsum("pi,...pq,qj->...ij", Cm, M, Cm, optimize=True)

    # ---- overlap, kinetic
    Sp = (np.pi / pp) ** 1.5 * Kab
    Tp = mu * (3.0 - 2.0 * mu * R2) * Sp
    dSp = -2.0 * mu[None] * np.moveaxis(AB, -1, 0) * Sp[None]                  # d/dA_p (first centre)
    dTp = mu[None] * (-4.0 * mu[None] * np.moveaxis(AB, -1, 0) * Sp[None] + (3.0 - 2.0 * mu * R2)[None] * dSp)

    # ---- nuclear attraction, its derivative w.r.t. the first centre and w.r.t. each nucleus
    Vp = np.zeros((Np, Np))
    dVp = np.zeros((3, Np, Np))
    dVop = np.zeros((A, 3, Np, Np))
    for c in range(A):
        PC = P - R[c][None, None, :]
        f0, f1 = boys01(pp * np.sum(PC * PC, axis=-1))
        pref = -Z[c] * (2.0 * np.pi / pp) * Kab
        Vp += pref * f0
        PCx = np.moveaxis(PC, -1, 0)
        dVp += pref[None] * (-2.0 * mu[None] * np.moveaxis(AB, -1, 0) * f0[None] - 2.0 * a[None, :, None] * PCx * f1[None])
        # d/dC F0(p |P-C|^2) = -F1 * 2 p (C - P)
        dVop[c] = pref[None] * (-f1[None]) * 2.0 * pp[None] * (-PCx)
    S = contract2(Sp)
    hcore = contract2(Tp + Vp)
    S = 0.5 * (S + S.T)
    hcore = 0.5 * (hcore + hcore.T)
    aoslices = np.stack([np.arange(A), np.arange(A) + 1], axis=1).astype(np.int64)

    # ---- nuclear repulsion
    enuc = 0.0
    gnuc = np.zeros((A, 3))
    for i in range(A):
        for j in range(A):
            if i == j:
                continue
            d = R[i] - R[j]
            r = np.linalg.norm(d)
            if j > i:
                enuc += Z[i] * Z[j] / r
            gnuc[i] -= Z[i] * Z[j] * d / r ** 3

    # ---- two-electron integrals, one contracted first index at a time
    q = pp.reshape(-1)
    Q = P.reshape(-1, 3)
    Kq = Kab.reshape(-1)
    eri = np.zeros((n, n, n, n))
    eri_ip1 = np.zeros((3, n, n, n, n)) if need_grad else np.zeros((3, 0))
    for m in range(n):
        I = np.nonzero(owner == m)[0]
        pI = pp[I][:, :, None]                               # (k,Np,1)
        PQ = P[I][:, :, None, :] - Q[None, None, :, :]       # (k,Np,Np^2,3)
        rho = pI * q[None, None, :] / (pI + q[None, None, :])
        f0, f1 = boys01(rho * np.sum(PQ * PQ, axis=-1))
        pref = (2.0 * np.pi ** 2.5 / (pI * q[None, None, :] * np.sqrt(pI + q[None, None, :]))
                * Kab[I][:, :, None] * Kq[None, None, :])
        w = cn[I][:, None, None]                             # weight of the first primitive

        def fold(X):                                         # (k,Np,Np^2) -> (n,n,n): sum p in m, contract q,r,s
            Y = np.tensordot(Cm, np.sum(w * X, axis=0), axes=(0, 0))            # (n, Np^2)
            Y = Y.reshape(n, Np, Np)
            return np.einsum("jrs,rk,sl->jkl", Y, Cm, Cm, optimize=True)

        eri[m] = fold(pref * f0)
        if need_grad:
            for x in range(3):
                d = pref * (-2.0 * mu[I][:, :, None] * AB[I][:, :, None, x] * f0
                            - 2.0 * rho * (a[I][:, None, None] / pI) * PQ[..., x] * f1)
                eri_ip1[x, m] = -fold(d)                     # electron gradient = - centre gradient

    if need_grad:
        ipovlp = -contract2(dSp)                             # <grad mu | nu>
        dH1 = contract2(dTp + dVp)                           # d/d(centre of mu) of hcore_{mu nu}, nuclei fixed
        dhcore = contract2(dVop)                             # (A,3,n,n) operator part
        for at in range(A):
            # the basis function of atom `at` moves with it: row `at` and, by symmetry, column `at`
            dhcore[at, :, at, :] += dH1[:, at, :]
            dhcore[at, :, :, at] += dH1[:, at, :]
    else:
        ipovlp = np.zeros((3, n, n))
        dhcore = np.zeros((A, 3, n, n))
    ne = int(round(float(np.sum(Z))))
    if nelec is None:
        nelec = ((ne + 1) // 2, ne // 2)
    return HChainMol(S=S, hcore=hcore, eri=eri, ipovlp=ipovlp, dhcore=dhcore, eri_ip1=eri_ip1,
                     aoslices=aoslices, enuc=float(enuc), gnuc=gnuc, integral_symmetry=True, coords=R, nelec=tuple(nelec), charges=Z,
                     exponents=tuple(float(x) for x in ex), coefficients=tuple(float(x) for x in co))


def hydrogen_chain(natm: int, spacing: float, need_grad: bool = True) -> HChainMol:
    """Equidistant linear H_n along x (``spacing`` in Bohr), STO-3G (``scripts/PES_H_chain``)."""
    coords = np.zeros((natm, 3))
    coords[:, 0] = spacing * np.arange(natm)
    return s_gaussian_mol(coords, need_grad=need_grad)
