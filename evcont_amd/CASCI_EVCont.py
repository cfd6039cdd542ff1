"""``CAS_EVCont_obj``: container for continuation training data from CASCI states (mirror of the
class in ``evcont/CASCI_EVCont.py:93-361``: constructor, attributes, ``prune_datapoints``).

Generating CASCI training states needs PySCF (RHF + CASCI) and pygnme (non-orthogonal Wick's theorem
for the transition RDMs between different orbital sets); that is training-time host work outside the
accelerated path (out of scope).  ``append_to_rdms`` therefore delegates the per-pair
overlap / transition-RDM evaluation to a user-supplied ``pair_rdm_fun(casci_bra, casci_ket) ->
(ovlp, rdm1, rdm2)`` in the OAO basis (what ``CASCI_EVCont.py:204-335`` computes with pygnme) and
raises ImportError when neither that nor PySCF is available.  Everything downstream — array growth,
pruning, the device-resident packed copy — is shared with the other containers.
"""
from __future__ import annotations

import numpy as np

from .containers import TRDMContainer


class CAS_EVCont_obj(TRDMContainer):
    def __init__(self, ncas, neleca, casci_solver=None, pair_rdm_fun=None):
        super().__init__()
        self.ncas = ncas
        self.neleca = neleca
        self.cascis = []
        self.casci_solver = casci_solver
        self.pair_rdm_fun = pair_rdm_fun

    def append_to_rdms(self, mol):
        if self.casci_solver is None:
            try:
                from pyscf.mcscf import CASCI
            except ImportError as e:
                raise ImportError("CAS_EVCont_obj.append_to_rdms needs PySCF (RHF/CASCI training states are "
                                  "generated on the host) or an explicit casci_solver") from e
            self.casci_solver = CASCI
        if self.pair_rdm_fun is None:
            raise ImportError("CAS_EVCont_obj.append_to_rdms needs pair_rdm_fun(casci_bra, casci_ket) -> "
                              "(overlap, rdm1, rdm2) in the OAO basis (the reference evaluates it with pygnme, "
                              "CASCI_EVCont.py:204-335)")
        mf = mol.copy().RHF()
        mf.kernel()
        assert mf.converged
        casci_bra = self.casci_solver(mf, self.ncas, self.neleca)
        casci_bra.kernel()
        assert casci_bra.fcisolver.converged
        self.cascis.append(casci_bra)
        T1 = len(self.cascis)
        n = casci_bra.mo_coeff.shape[0]
        ovlp, one, two = np.empty(T1), np.empty((T1, n, n)), np.empty((T1, n, n, n, n))
        for i, ket in enumerate(self.cascis):
            ovlp[i], one[i], two[i] = self.pair_rdm_fun(casci_bra, ket)
        self._append_state(ovlp, one, two)

    def prune_datapoints(self, keep_ids):
        """``CASCI_EVCont.py:345-361``."""
        self._prune_arrays(keep_ids)
        self.cascis = [self.cascis[i] for i in keep_ids]
