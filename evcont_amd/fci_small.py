"""A small determinant full-CI solver with the two calls the FCI training-state container makes
on a PySCF ``cisolver`` (``FCI_EVCont.py:72-74,118-121``):

    e, ci = solver.kernel(h1, h2, norb, nelec, nroots=k)
    dm1, dm2 = solver.trans_rdm12(cibra, ciket, norb, nelec)

so that hydrogen-chain training data (BASELINE configs 0-1) can be generated without PySCF
(SURVEY.md §8f-2).  Conventions follow ``pyscf.fci.direct_spin1``: CI vectors are ``(na, nb)`` arrays
over alpha/beta occupation strings ordered by their integer value; ``dm1[p,q] = <q^+ p>`` and
``dm2[p,q,r,s] = <p^+ r^+ s q>`` (chemists' order, spin summed), so that
``E = h1:dm1 + 1/2 h2:dm2`` with ``h2`` in chemists' notation.

Host code for up to ~12 orbitals (training-state generation is outside the accelerated hot path).
"""
from __future__ import annotations

from itertools import combinations
from typing import List, Tuple

import numpy as np
import scipy.sparse as sp
from scipy.sparse.linalg import LinearOperator, eigsh


def _strings(norb: int, nocc: int) -> List[int]:
    return sorted(sum(1 << i for i in occ) for occ in combinations(range(norb), nocc))


def _excitation_ops(norb: int, nocc: int):
    """E[p][q] = a_p^+ a_q on the occupation strings, as CSR matrices (rows = resulting string)."""
    strs = _strings(norb, nocc)
    index = {s: i for i, s in enumerate(strs)}
    ns = len(strs)
    ops = [[None] * norb for _ in range(norb)]
    for p in range(norb):
        for q in range(norb):
            rows, cols, vals = [], [], []
            for j, s in enumerate(strs):
                if not (s >> q) & 1:
                    continue
                if p == q:
                    rows.append(j); cols.append(j); vals.append(1.0)
                    continue
                if (s >> p) & 1:
                    continue
                t = (s ^ (1 << q)) | (1 << p)
                lo, hi = (p, q) if p < q else (q, p)
                between = bin(s & (((1 << hi) - 1) ^ ((1 << (lo + 1)) - 1))).count("1")
                rows.append(index[t]); cols.append(j); vals.append(-1.0 if between & 1 else 1.0)
            ops[p][q] = sp.csr_matrix((vals, (rows, cols)), shape=(ns, ns))
    return ops, ns


class SmallFCI:
    """Spin-adapted only through Ms: ``nelec = (nalpha, nbeta)``."""

    def __init__(self, tol: float = 1e-13, dense_limit: int = 1500):
        self.tol = tol
        self.dense_limit = dense_limit
        self._cache = {}
        self.converged = True

    def _ops(self, norb: int, nelec: Tuple[int, int]):
        key = (norb, tuple(nelec))
        if key not in self._cache:
            ea, na = _excitation_ops(norb, nelec[0])
            eb, nb = (ea, na) if nelec[1] == nelec[0] else _excitation_ops(norb, nelec[1])
            ebT = [[eb[p][q].T.tocsr() for q in range(norb)] for p in range(norb)]
            self._cache[key] = (ea, ebT, na, nb)
        return self._cache[key]

    def _excite_all(self, c, norb, nelec):
        """D[p*norb+q] = E_pq c for all p, q; shape (norb^2, na, nb)."""
        ea, ebT, na, nb = self._ops(norb, nelec)
        c = np.asarray(c, dtype=np.float64).reshape(na, nb)
        D = np.empty((norb * norb, na, nb))
        for p in range(norb):
            for q in range(norb):
                D[p * norb + q] = ea[p][q] @ c + (ebT[p][q].T @ c.T).T
        return D

    def contract(self, h1, h2, c, norb, nelec):
        """sigma = H c with H = sum h'_pq E_pq + 1/2 sum (pq|rs) E_pq E_rs, h'_pq = h_pq - 1/2 sum_r (pr|rq)."""
        ea, ebT, na, nb = self._ops(norb, nelec)
        h2 = np.asarray(h2, dtype=np.float64).reshape(norb, norb, norb, norb)
        hp = np.asarray(h1, dtype=np.float64) - 0.5 * np.einsum("prrq->pq", h2)
        D = self._excite_all(c, norb, nelec)
        sigma = np.tensordot(hp.reshape(-1), D, axes=(0, 0))
        G = (h2.reshape(norb * norb, norb * norb) @ D.reshape(norb * norb, -1)).reshape(D.shape)
        for p in range(norb):
            for q in range(norb):
                g = G[p * norb + q]
                sigma += 0.5 * (ea[p][q] @ g + (ebT[p][q].T @ g.T).T)
        return sigma

    def kernel(self, h1, h2, norb, nelec, nroots: int = 1, **_):
        """Lowest ``nroots`` eigenpairs.  Like PySCF: scalars/array for nroots == 1, lists otherwise."""
        if isinstance(nelec, (int, np.integer)):
            nelec = ((int(nelec) + 1) // 2, int(nelec) // 2)
        nelec = (int(nelec[0]), int(nelec[1]))
        _, _, na, nb = self._ops(norb, nelec)
        dim = na * nb
        mv = lambda v: self.contract(h1, h2, v.reshape(na, nb), norb, nelec).reshape(-1)
        if dim <= self.dense_limit:
            H = np.empty((dim, dim))
            eye = np.zeros(dim)
            for k in range(dim):
                eye[k] = 1.0
                H[:, k] = mv(eye)
                eye[k] = 0.0
            H = 0.5 * (H + H.T)
            w, v = np.linalg.eigh(H)
        else:
            op = LinearOperator((dim, dim), matvec=mv, dtype=np.float64)
            rng = np.random.default_rng(0)
            w, v = eigsh(op, k=max(nroots, 1), which="SA", tol=self.tol, v0=rng.standard_normal(dim),
                         ncv=max(20, 2 * nroots + 10))
            order = np.argsort(w)
            w, v = w[order], v[:, order]
        vecs = []
        for k in range(nroots):
            x = v[:, k].copy()
            x *= np.sign(x[np.argmax(np.abs(x))])     # fixed sign convention
            vecs.append(x.reshape(na, nb))
        if nroots == 1:
            return float(w[0]), vecs[0]
        return [float(x) for x in w[:nroots]], vecs

    def trans_rdm12(self, cibra, ciket, norb, nelec):
        if isinstance(nelec, (int, np.integer)):
            nelec = ((int(nelec) + 1) // 2, int(nelec) // 2)
        nelec = (int(nelec[0]), int(nelec[1]))
        n2 = norb * norb
        Dk = self._excite_all(ciket, norb, nelec).reshape(n2, -1)
        Db = self._excite_all(cibra, norb, nelec).reshape(n2, -1)
        bra = np.asarray(cibra, dtype=np.float64).reshape(-1)
        g1 = (Dk @ bra).reshape(norb, norb)                     # <bra| E_pq |ket>
        # <bra| E_pq E_rs |ket> = <E_qp bra | E_rs ket>
        DbT = Db.reshape(norb, norb, -1).transpose(1, 0, 2).reshape(n2, -1)
        M = (DbT @ Dk.T).reshape(norb, norb, norb, norb)
        dm2 = M - np.einsum("qr,ps->pqrs", np.eye(norb), g1)   # <p^+ r^+ s q>
        return g1.T.copy(), dm2                                 # dm1[p,q] = <q^+ p>

    def make_rdm12(self, ci, norb, nelec):
        return self.trans_rdm12(ci, ci, norb, nelec)

    def energy(self, h1, h2, ci, norb, nelec) -> float:
        dm1, dm2 = self.make_rdm12(ci, norb, nelec)
        return float(np.sum(np.asarray(h1) * dm1.T) + 0.5 * np.sum(np.asarray(h2).reshape(dm2.shape) * dm2))
