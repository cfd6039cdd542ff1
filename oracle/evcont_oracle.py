"""CPU oracle for the eigenvector-continuation energy/force hot path.

TEST INFRASTRUCTURE ONLY.  This file is a numpy restatement of the algorithm of
the reference (BoothGroup/evcont) hot path, at *array level* (no ``mol`` object:
the AO integrals are handed over in an :class:`AOBundle`).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product package ``evcont_amd`` never does.

Pinning: every function here is checked in ``tests/test_oracle_golden.py``
against golden vectors in ``tests/golden/*.npz`` that were produced by running
the *reference's own functions* in the build container
(``tests/golden/make_golden.py``; the reference's modules import unmodified once
an empty ``pyscf`` namespace is present, and a fake ``mol`` serves seeded AO
arrays).  What is NOT pinned is the PySCF boundary itself (libcint integrals,
``ao2mo.kernel``): the reference pins no PySCF version and holds no tests, so
agreement "on identical PySCF inputs" means "on identical AO arrays".

Reference citations are ``file:line`` below ``/root/reference/evcont``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np
import scipy.linalg as sla


# --------------------------------------------------------------------------
# Array-level stand-in for the PySCF ``mol`` the reference queries.
# --------------------------------------------------------------------------
@dataclass
class AOBundle:
    """AO-basis arrays of ONE geometry (what the reference pulls out of PySCF).

    ``S``        int1e_ovlp                      (N,N)    ab_initio_gradients_loewdin.py:336
    ``hcore``    scf.hf.get_hcore(mol)           (N,N)    ab_initio_gradients_loewdin.py:338
    ``eri``      int2e, chemists' (ab|cd), s1    (N,N,N,N) ab_initio_gradients_loewdin.py:283
    ``ipovlp``   int1e_ipovlp, comp=3            (3,N,N)  ab_initio_gradients_loewdin.py:25
    ``dhcore``   hcore_generator()(atom)         (A,3,N,N) ab_initio_gradients_loewdin.py:147-149
    ``eri_ip1``  int2e_ip1, comp=3               (3,N,N,N,N) ab_initio_gradients_loewdin.py:284
    ``aoslices`` aoslice_by_atom()[:, 2:4]       (A,2) int
    ``enuc``     mol.energy_nuc()                float
    ``gnuc``     grad.RHF(...).grad_nuc()        (A,3)
    """

    S: np.ndarray
    hcore: np.ndarray
    eri: np.ndarray
    ipovlp: np.ndarray
    dhcore: np.ndarray
    eri_ip1: np.ndarray
    aoslices: np.ndarray
    enuc: float
    gnuc: np.ndarray

    @property
    def nao(self) -> int:
        return int(self.S.shape[0])

    @property
    def natm(self) -> int:
        return int(self.aoslices.shape[0])


# --------------------------------------------------------------------------
# electron_integral_utils.py
# --------------------------------------------------------------------------
def loewdin_trafo(S_ao: np.ndarray) -> np.ndarray:
    """X = S^(-1/2) with the 1e-15 eigenvalue guard (electron_integral_utils.py:6-18)."""
    s, U = np.linalg.eigh(S_ao)
    f = np.zeros_like(s)
    ok = s > 1.0e-15
    f[ok] = 1.0 / np.sqrt(s[ok])
    return (U * f) @ U.conj().T


def pack_pair_sym(h2: np.ndarray, diag_multiplier: float = 1.0) -> np.ndarray:
    """(N,N,N,N) -> row-major lower triangle of the (N^2,N^2) matrix, diagonal scaled.

    Out-of-place equivalent of electron_integral_utils.py:38-66 (the reference
    scales the diagonal in place and restores it afterwards)."""
    n = h2.shape[0]
    assert h2.shape == (n, n, n, n)
    mat = np.array(h2, dtype=np.float64).reshape(n * n, n * n)
    r, c = np.tril_indices(n * n)
    out = mat[r, c].copy()
    out[r == c] *= diag_multiplier
    return out


def unpack_pair_sym(v: np.ndarray, norb: int) -> np.ndarray:
    """Inverse of :func:`pack_pair_sym` with multiplier 1 (electron_integral_utils.py:69-88)."""
    n2 = norb * norb
    mat = np.zeros((n2, n2))
    r, c = np.tril_indices(n2)
    mat[r, c] = v
    mat[c, r] = v
    return mat.reshape(norb, norb, norb, norb)


def integrals_oao(b: AOBundle, X: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
    """h1 = X^T hcore X ; h2 = (X x X x X x X) eri  (electron_integral_utils.py:122-138).

    The reference delegates the 4-index step to ``pyscf.ao2mo.kernel``; its
    published definition is the dense transformation restated here."""
    if X is None:
        X = loewdin_trafo(b.S)
    h1 = X.T @ b.hcore @ X
    t = np.tensordot(b.eri, X, axes=([3], [0]))       # (a,b,c,l)
    t = np.tensordot(t, X, axes=([2], [0]))           # (a,b,l,k)
    t = np.tensordot(t, X, axes=([1], [0]))           # (a,l,k,j)
    t = np.tensordot(t, X, axes=([0], [0]))           # (l,k,j,i)
    return h1, np.ascontiguousarray(t.transpose(3, 2, 1, 0))


# --------------------------------------------------------------------------
# ab_initio_eigenvector_continuation.py
# --------------------------------------------------------------------------
def subspace_hamiltonian(h1, h2, one_RDM, two_RDM, hermitian=True) -> np.ndarray:
    """H_ab for the four t-RDM layouts (ab_initio_eigenvector_continuation.py:38-71).

    Note the reference's quirk, reproduced: for the pair-compressed layouts in
    Hermitian mode only the LOWER triangle receives the two-body part."""
    T = one_RDM.shape[0]
    H = np.tensordot(one_RDM, h1, axes=2).astype(np.float64)
    nd = two_RDM.ndim
    lo = np.tril_indices(T)
    up = np.triu_indices(T)
    if nd == 6:
        H = H + 0.5 * np.tensordot(two_RDM, h2, axes=4)
    elif nd == 5:
        H[lo] += 0.5 * np.tensordot(two_RDM, h2, axes=4)
        if not hermitian:
            H[up] = H.T.conj()[up]
    elif nd == 3:
        H = H + two_RDM @ pack_pair_sym(h2, 0.5)
    elif nd == 2:
        H[lo] += two_RDM @ pack_pair_sym(h2, 0.5)
        if not hermitian:
            H[up] = H.T.conj()[up]
    else:
        raise AssertionError("two_RDM must have 2, 3, 5 or 6 dimensions")
    return H


def _gen_eig(H, S, hermitian):
    """eigh(H,S) (lower triangles) or eig(H,S) + |Im|<1e-5 filter
    (ab_initio_eigenvector_continuation.py:73-81)."""
    if hermitian:
        vals, vecs = sla.eigh(H, S)
    else:
        vals, vecs = sla.eig(H, S)
    keep = np.abs(vals.imag) < 1.0e-5
    return vals[keep], vecs[:, keep]


def approximate_ground_state(h1, h2, one_RDM, two_RDM, S, hermitian=True):
    """ab_initio_eigenvector_continuation.py:12-90."""
    H = subspace_hamiltonian(h1, h2, one_RDM, two_RDM, hermitian)
    vals, vecs = _gen_eig(H, S, hermitian)
    k = int(np.argmin(vals.real))
    return float(vals[k].real), np.array(vecs[:, k].real)


def approximate_multistate(h1, h2, one_RDM, two_RDM, S, nroots=1, hermitian=True):
    """ab_initio_eigenvector_continuation.py:93-175."""
    H = subspace_hamiltonian(h1, h2, one_RDM, two_RDM, hermitian)
    vals, vecs = _gen_eig(H, S, hermitian)
    assert vals.shape[0] >= nroots
    order = np.argsort(vals.real)[:nroots]
    return np.array(vals[order].real), np.array(vecs[:, order].real.T)


def approximate_ground_state_OAO(b: AOBundle, one_RDM, two_RDM, S, hermitian=True):
    """ab_initio_eigenvector_continuation.py:178-211."""
    h1, h2 = integrals_oao(b)
    e, c = approximate_ground_state(h1, h2, one_RDM, two_RDM, S, hermitian)
    return e + b.enuc, c


def approximate_multistate_OAO(b: AOBundle, one_RDM, two_RDM, S, nroots=1, hermitian=True):
    """ab_initio_eigenvector_continuation.py:214-250."""
    h1, h2 = integrals_oao(b)
    e, c = approximate_multistate(h1, h2, one_RDM, two_RDM, S, nroots, hermitian)
    return e + b.enuc, c


# --------------------------------------------------------------------------
# ab_initio_gradients_loewdin.py
# --------------------------------------------------------------------------
def overlap_grad(ipovlp: np.ndarray, aoslices: np.ndarray) -> np.ndarray:
    """dS[mu,nu,A,x] = -<grad mu|nu>[mu in A] - <grad nu|mu>[nu in A]
    (ab_initio_gradients_loewdin.py:13-38)."""
    n = ipovlp.shape[1]
    A = len(aoslices)
    d = np.zeros((3, A, n, n))
    for a, (p0, p1) in enumerate(aoslices):
        d[:, a, p0:p1, :] = -ipovlp[:, p0:p1, :]
    d = d + d.transpose(0, 1, 3, 2)
    return np.ascontiguousarray(d.transpose(2, 3, 1, 0))


def loewdin_trafo_grad_bucketed(S_ao: np.ndarray) -> np.ndarray:
    """Derivative of X=S^(-1/2) w.r.t. the symmetrised unit perturbations
    0.5(e_a e_b^T + e_b e_a^T), following the reference's degenerate
    perturbation theory: eigenvalues equal after rounding to 5 decimals form a
    bucket; inside a bucket the eigenvectors are rotated to diagonalise the
    projected perturbation and do not mix (ab_initio_gradients_loewdin.py:41-112).

    Returned as R[p,q,a,b] = d X_pq / d S_ab^(sym); it is symmetric under
    (p,q)<->(a,b), which is what lets the reference contract its first index
    pair with dS/dR in get_derivative_ao_mo_trafo (:128-132).

    Written as an explicit loop over the N^2 perturbations (small cases only)."""
    s, V = np.linalg.eigh(S_ao)
    n = s.shape[0]
    rs = np.round(s, decimals=5)
    buckets = [np.flatnonzero(rs == v) for v in np.unique(rs)]
    same = np.zeros((n, n), dtype=bool)
    for ids in buckets:
        same[np.ix_(ids, ids)] = True
    gap = s[None, :] - s[:, None]                    # gap[j,k] = s_k - s_j
    f = np.where(s > 1.0e-15, 1.0 / np.sqrt(s), 0.0)
    fp = np.where(s > 1.0e-15, -(0.5 / np.sqrt(s) ** 3), 0.0)
    out = np.zeros((n, n, n, n))
    for a in range(n):
        for b in range(n):
            W = V.copy()
            for ids in buckets:
                sub = V[:, ids]
                proj = 0.5 * (np.outer(sub[a], sub[b]) + np.outer(sub[b], sub[a]))
                _, rot = np.linalg.eigh(proj)
                W[:, ids] = sub @ rot
            pert = 0.5 * (np.outer(W[a], W[b]) + np.outer(W[b], W[a]))   # W^T P_ab W
            Z = np.zeros((n, n))
            Z[~same] = pert[~same] / gap[~same]
            dW = W @ Z
            ds = np.diag(pert)
            dX = (dW * f) @ W.T + (W * (fp * ds)) @ W.T + (W * f) @ dW.T
            out[:, :, a, b] = dX
    return out


def loewdin_trafo_grad_dk(S_ao: np.ndarray) -> np.ndarray:
    """Closed-form (Daleckii-Krein divided-difference) derivative of S^(-1/2):
    dX = U [ (U^T dS U) o F ] U^T,  F_ij = -1/(sqrt(s_i) sqrt(s_j) (sqrt(s_i)+sqrt(s_j))).

    This is what the GPU path evaluates (in adjoint form); it coincides with
    :func:`loewdin_trafo_grad_bucketed` for non-degenerate and exactly
    degenerate spectra and differs by O(bucket width) inside a bucket."""
    s, U = np.linalg.eigh(S_ao)
    F = loewdin_divided_differences(s)
    n = s.shape[0]
    # R[p,q,a,b] = sum_ij U_pi U_qj F_ij 0.5 (U_ai U_bj + U_bi U_aj)
    t = np.einsum("pi,qj,ij,ai,bj->pqab", U, U, F, U, U, optimize=True)
    return 0.5 * (t + t.transpose(0, 1, 3, 2)).reshape(n, n, n, n)


def loewdin_divided_differences(s: np.ndarray) -> np.ndarray:
    """F_ij = (f(s_i)-f(s_j))/(s_i-s_j) for f(s)=s^(-1/2) with the 1e-15 guard."""
    n = s.shape[0]
    ok = s > 1.0e-15
    r = np.where(ok, np.sqrt(np.where(ok, s, 1.0)), 0.0)
    f = np.where(ok, 1.0 / np.where(ok, r, 1.0), 0.0)
    F = np.zeros((n, n))
    for i in range(n):
        for j in range(n):
            if ok[i] and ok[j]:
                F[i, j] = -1.0 / (r[i] * r[j] * (r[i] + r[j]))
            elif i != j and s[i] != s[j]:
                F[i, j] = (f[i] - f[j]) / (s[i] - s[j])
    return F


def derivative_ao_mo_trafo(b: AOBundle, bucketed: bool = True) -> np.ndarray:
    """dX[k,l,A,x] (ab_initio_gradients_loewdin.py:115-134)."""
    lg = loewdin_trafo_grad_bucketed(b.S) if bucketed else loewdin_trafo_grad_dk(b.S)
    dS = overlap_grad(b.ipovlp, b.aoslices)
    return np.tensordot(lg, dS, axes=([0, 1], [0, 1]))


def one_el_grad_ao(b: AOBundle) -> np.ndarray:
    """(A,3,N,N) -> (N,N,A,3)  (ab_initio_gradients_loewdin.py:137-152)."""
    return np.ascontiguousarray(np.asarray(b.dhcore).transpose(2, 3, 0, 1))


def one_el_grad(b: AOBundle, X=None, dX=None) -> np.ndarray:
    """d h1^OAO / dR = dX^T h X + X^T h dX + X^T dh X  -> (N,N,A,3)
    (ab_initio_gradients_loewdin.py:155-187)."""
    if X is None:
        X = loewdin_trafo(b.S)
    if dX is None:
        dX = derivative_ao_mo_trafo(b)
    hX = b.hcore @ X                                           # (i,n)
    g = np.einsum("ijkl,in->jnkl", dX, hX)
    g = g + g.swapaxes(0, 1)
    g = g + np.einsum("ij,iklm,kn->jnlm", X, one_el_grad_ao(b), X, optimize=True)
    return g


def two_el_grad(h2_ao, two_rdm, X, dX, ip1, atm_slices) -> np.ndarray:
    """Two-electron gradient (A,3)  (ab_initio_gradients_loewdin.py:190-252).

    T1[A,x] = sum Gs_ijkl (ab|cd) dX_ai[A,x] X_bj X_ck X_dl,
              Gs = G + G^(1,0,2,3) + G^(3,2,1,0) + G^(2,3,0,1)          (:210-222)
    G^AO   = (X x X x X x X) G                                           (:224-232)
    T2[x,m,a] = sum_bcd ip1[x,m,b,c,d] Gs^AO[a,b,c,d],
              Gs^AO = G^AO + ^(1,0,3,2) + ^(2,3,0,1) + ^(3,2,1,0)        (:234-242)
    out[A,x] = T1[A,x] - sum_{m in A} T2[x,m,m]                          (:244-252)
    """
    n = two_rdm.shape[0]
    Gs = (two_rdm + two_rdm.transpose(1, 0, 2, 3) + two_rdm.transpose(3, 2, 1, 0)
          + two_rdm.transpose(2, 3, 0, 1))
    # three-quarter transformed ERI K[a,j,k,l]
    K = np.tensordot(h2_ao, X, axes=([3], [0]))        # a b c l
    K = np.tensordot(K, X, axes=([2], [0]))            # a b l k
    K = np.tensordot(K, X, axes=([1], [0]))            # a l k j
    K = K.transpose(0, 3, 2, 1)                        # a j k l
    Y = np.tensordot(K, Gs, axes=([1, 2, 3], [1, 2, 3]))   # Y[a,i]
    t1 = np.tensordot(Y, dX, axes=([0, 1], [0, 1]))        # (A,3)

    G = np.tensordot(X, two_rdm, axes=([1], [0]))          # a j k l
    G = np.tensordot(X, G, axes=([1], [1]))                # b a k l
    G = np.tensordot(X, G, axes=([1], [2]))                # c b a l
    G = np.tensordot(X, G, axes=([1], [3]))                # d c b a
    G = G.transpose(3, 2, 1, 0)
    Gs_ao = G + G.transpose(1, 0, 3, 2) + G.transpose(2, 3, 0, 1) + G.transpose(3, 2, 1, 0)
    t2 = np.tensordot(ip1.reshape(3, n, n ** 3), Gs_ao.reshape(n, n ** 3), axes=([2], [1]))  # x,m,a
    out = t1.copy()
    for A, (p0, p1) in enumerate(atm_slices):
        for m in range(p0, p1):
            out[A, :] -= t2[:, m, m]
    return out


def grad_elec_OAO(b: AOBundle, one_rdm, two_rdm, X=None, dX=None, bucketed=True) -> np.ndarray:
    """ab_initio_gradients_loewdin.py:255-305."""
    if X is None:
        X = loewdin_trafo(b.S)
    if dX is None:
        dX = derivative_ao_mo_trafo(b, bucketed=bucketed)
    h1_jac = one_el_grad(b, X, dX)
    g2 = two_el_grad(b.eri, two_rdm, X, dX, b.eri_ip1, [tuple(s) for s in b.aoslices])
    return np.tensordot(one_rdm, h1_jac, axes=([0, 1], [0, 1])) + 0.5 * g2


def pair_weights(c: np.ndarray) -> np.ndarray:
    """w_p for p=(a>=b): 2 c_a c_b off-diagonal, c_a^2 diagonal
    (ab_initio_gradients_loewdin.py:345-353)."""
    m = 2.0 * np.outer(c, c)
    m[np.diag_indices_from(m)] *= 0.5
    return m[np.tril_indices(len(c))]


def predicted_rdms(c, one_RDM, two_RDM, norb):
    """D_pred, Gamma_pred for the four layouts (ab_initio_gradients_loewdin.py:343-361)."""
    cc = np.outer(c, c)
    D = np.tensordot(cc, one_RDM, axes=2)
    if two_RDM.ndim in (2, 5):
        G = np.tensordot(pair_weights(c), two_RDM, axes=1)
    else:
        G = np.tensordot(cc, two_RDM, axes=2)
    if G.ndim != 4:
        G = unpack_pair_sym(G, norb)
    return D, G


def energy_with_grad(b: AOBundle, one_RDM, two_RDM, S, hermitian=True,
                     return_density_matrices=False, bucketed=True):
    """ab_initio_gradients_loewdin.py:308-379."""
    X = loewdin_trafo(b.S)
    h1, h2 = integrals_oao(b, X)
    e, c = approximate_ground_state(h1, h2, one_RDM, two_RDM, S, hermitian)
    D, G = predicted_rdms(c, one_RDM, two_RDM, b.nao)
    g = grad_elec_OAO(b, D, G, X=X, bucketed=bucketed) + b.gnuc
    if return_density_matrices:
        return e + b.enuc, g, D, G
    return e + b.enuc, g


# --------------------------------------------------------------------------
# t-RDM container growth / pruning semantics (FCI_EVCont.py:106-151)
# --------------------------------------------------------------------------
def grow_trdms(overlap, one_rdm, two_rdm, new_ovlp_row: Sequence[float],
               new_rdm1: Sequence[np.ndarray], new_rdm2: Sequence[np.ndarray]):
    """Append one training state: new row AND column get the same (untransposed)
    arrays, as FCI_EVCont.py:118-127 does."""
    T = len(new_ovlp_row)
    n = new_rdm1[0].shape[0]
    o = np.ones((T, T))
    d = np.ones((T, T, n, n))
    g = np.ones((T, T, n, n, n, n))
    if overlap is not None:
        o[:-1, :-1] = overlap
        d[:-1, :-1] = one_rdm
        g[:-1, :-1] = two_rdm
    for i in range(T):
        o[-1, i] = o[i, -1] = new_ovlp_row[i]
        d[-1, i] = d[i, -1] = new_rdm1[i]
        g[-1, i] = g[i, -1] = new_rdm2[i]
    return o, d, g


def prune_trdms(overlap, one_rdm, two_rdm, keep_ids):
    """FCI_EVCont.py:133-151."""
    ix = np.ix_(keep_ids, keep_ids)
    return overlap[ix], one_rdm[ix], two_rdm[ix]
