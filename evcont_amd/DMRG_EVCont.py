"""``DMRG_EVCont_obj``: container for continuation training data from matrix-product states (mirror of
the class in ``evcont/DMRG_EVCont.py:429-496``: constructor, attributes, ``append_to_rdms``,
``prune_datapoints``).

The MPS optimisation and the MPS-MPS transition RDMs are block2 host work outside the accelerated path
(out of scope); ``append_method`` has the reference's signature
``(mols, tags, overlap=, one_rdm=, two_rdm=, converge_dmrg_fun=, mem=) -> (overlap, one_rdm, two_rdm)``
(``DMRG_EVCont.py:21-87``) and must be supplied by the caller when block2 is not installed.  The arrays
it returns use the chemists' ordering Gamma_pqrs <-> (pq|rs), i.e. block2's 2-PDM transposed with
``(0,3,1,2)`` (``DMRG_EVCont.py:78``).  Scripts may also assign ``np.load`` results to
``overlap/one_rdm/two_rdm`` directly (``md_H30_evcont_from_DMRG.py:72-85``); the device copy follows.
"""
from __future__ import annotations

from .containers import TRDMContainer


def _missing(*_a, **_k):
    raise ImportError("DMRG_EVCont_obj needs an append_method (block2-based in the reference, "
                      "DMRG_EVCont.py:21-427); block2 is not part of this build")


class DMRG_EVCont_obj(TRDMContainer):
    def __init__(self, dmrg_converge_fun=None, append_method=None, mem=5):
        super().__init__()
        self.solver = dmrg_converge_fun
        self.append_method = append_method if append_method is not None else _missing
        self.mols = []
        self.tags = []
        self.max_tag = 0
        self.mem = mem

    def append_to_rdms(self, mol):
        self.mols.append(mol)
        self.tags.append(self.max_tag)
        self.max_tag += 1
        self.overlap, self.one_rdm, self.two_rdm = self.append_method(
            self.mols, self.tags, overlap=self.overlap, one_rdm=self.one_rdm, two_rdm=self.two_rdm,
            converge_dmrg_fun=self.solver, mem=self.mem)

    def prune_datapoints(self, keep_ids):
        self._prune_arrays(keep_ids)
        self.mols = [self.mols[i] for i in keep_ids]
        self.tags = [self.tags[i] for i in keep_ids]
