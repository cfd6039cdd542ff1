"""``FCI_EVCont_obj``: continuation training data from full-CI states (mirror of
``evcont/FCI_EVCont.py``: same constructor, attributes and methods).

The FCI solve itself is host work outside the accelerated path; the integrals it consumes come from
``evcont_amd.electron_integral_utils`` (four-index transform on the GPU).  ``cisolver`` is any object with
PySCF's ``kernel(h1, h2, norb, nelec, nroots=)`` / ``trans_rdm12(bra, ket, norb, nelec)`` interface; the
default is ``pyscf.fci.direct_spin0.FCI()`` when PySCF is importable and ``fci_small.SmallFCI`` otherwise.
"""
from __future__ import annotations

import numpy as np

from .containers import TRDMContainer
from .electron_integral_utils import get_basis, get_integrals
from .integrals import is_array_mol, nao_of, energy_nuc


def _default_cisolver():
    try:
        from pyscf import fci
        return fci.direct_spin0.FCI()
    except ImportError:
        from .fci_small import SmallFCI
        return SmallFCI()


class FCI_EVCont_obj(TRDMContainer):
    """Holds ``fcivecs``, ``ens``, ``mol_index`` and the t-RDMs ``overlap/one_rdm/two_rdm``
    (``FCI_EVCont.py:10-56``)."""

    def __init__(self, cisolver=None, cibasis="canonical", nroots=1, roots_train=None):
        super().__init__()
        self.cisolver = cisolver if cisolver is not None else _default_cisolver()
        self.cibasis = cibasis
        self.nroots = nroots
        if roots_train is None:
            self.roots_train = list(range(nroots))
        else:
            assert isinstance(roots_train, list)
            self.roots_train = roots_train
        self.fcivecs = []
        self.ens = []
        self.mol_index = []

    def append_to_rdms(self, mol):
        """Solve FCI at ``mol`` and grow the t-RDMs by its trained roots (``FCI_EVCont.py:58-131``)."""
        basis = get_basis(mol, basis_type=self.cibasis)
        h1, h2 = get_integrals(mol, basis)
        n = nao_of(mol)
        nroots_train = max(self.roots_train) + 1
        e_all, fcivec_all = self.cisolver.kernel(h1, h2, n, mol.nelec, nroots=nroots_train)
        if nroots_train == 1:
            e_all, fcivec_all = [e_all], [fcivec_all]
        if self.cibasis != "OAO":
            # rotate the CI vectors from the solver's basis into the OAO basis (:80-87)
            from pyscf.fci.addons import transform_ci
            S = mol.S if is_array_mol(mol) else mol.intor("int1e_ovlp")
            u = np.einsum("ji,jk,kl->il", basis, S, get_basis(mol))
            fcivec_all = [transform_ci(v, mol.nelec, u) for v in fcivec_all]
        mindex = 0 if len(self.mol_index) == 0 else max(self.mol_index) + 1
        for ind in range(len(e_all)):
            if ind not in self.roots_train:
                continue
            fcivec = fcivec_all[ind]
            self.fcivecs.append(fcivec)
            self.ens.append(e_all[ind] + energy_nuc(mol))
            self.mol_index.append(mindex)
            T1 = len(self.fcivecs)
            ovlp = np.empty(T1)
            one = np.empty((T1, n, n))
            two = np.empty((T1, n, n, n, n))
            for i in range(T1):
                ovlp[i] = np.dot(np.ravel(self.fcivecs[-1]).conj(), np.ravel(self.fcivecs[i]))
                one[i], two[i] = self.cisolver.trans_rdm12(self.fcivecs[-1], self.fcivecs[i], n, mol.nelec)
            self._append_state(ovlp, one, two)

    def prune_datapoints(self, keep_ids):
        """Keep the listed training states (``FCI_EVCont.py:133-151``)."""
        self._prune_arrays(keep_ids)
        self.fcivecs = [self.fcivecs[i] for i in keep_ids]
        self.ens = [self.ens[i] for i in keep_ids]
