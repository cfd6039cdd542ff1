"""CPU tests of the training-data containers (evcont_amd/containers.py and the three mirror classes):
growth/prune array semantics of the reference (FCI_EVCont.py:106-151, DMRG_EVCont.py:451-496) and the
attribute surface the scripts rely on.  The FCI container is exercised end to end on the GPU
(tests/test_gpu_hchain.py)."""
import numpy as np
import pytest

from evcont_amd.containers import TRDMContainer, grow_trdms
from oracle import evcont_oracle as orc


def test_grow_matches_reference_semantics():
    rng = np.random.default_rng(0)
    n = 3
    S = one = two = None
    So = oo = to = None
    for T1 in range(1, 5):
        ov = rng.standard_normal(T1)
        r1 = rng.standard_normal((T1, n, n))
        r2 = rng.standard_normal((T1, n, n, n, n))
        S, one, two = grow_trdms(S, one, two, ov, r1, r2)
        So, oo, to = orc.grow_trdms(So, oo, to, list(ov), list(r1), list(r2))
        assert np.array_equal(S, So) and np.array_equal(one, oo) and np.array_equal(two, to)
        # new row and new column carry the SAME (untransposed) blocks
        assert np.array_equal(one[-1, :], one[:, -1]) and np.array_equal(two[-1, :], two[:, -1])
    c = TRDMContainer()
    c.overlap, c.one_rdm, c.two_rdm = S, one, two
    c.prune_datapoints([3, 0])
    So, oo, to = orc.prune_trdms(S, one, two, [3, 0])
    assert np.array_equal(c.overlap, So) and np.array_equal(c.one_rdm, oo) and np.array_equal(c.two_rdm, to)
    assert c.ntrain == 2


def test_dmrg_container_surface():
    from evcont_amd.DMRG_EVCont import DMRG_EVCont_obj
    from evcont_amd.CASCI_EVCont import CAS_EVCont_obj
    calls = []

    def fake_append(mols, tags, overlap=None, one_rdm=None, two_rdm=None, converge_dmrg_fun=None, mem=5):
        calls.append((len(mols), list(tags), mem))
        T, n = len(mols), 3
        return np.eye(T), np.zeros((T, T, n, n)), np.zeros((T, T, n, n, n, n))

    c = DMRG_EVCont_obj(dmrg_converge_fun="solver", append_method=fake_append, mem=7)
    assert (c.mols, c.tags, c.max_tag, c.overlap, c.mem, c.solver) == ([], [], 0, None, 7, "solver")
    c.append_to_rdms("molA")
    c.append_to_rdms("molB")
    c.append_to_rdms("molC")
    assert calls[-1] == (3, [0, 1, 2], 7) and c.max_tag == 3
    c.prune_datapoints([0, 2])
    assert c.mols == ["molA", "molC"] and c.tags == [0, 2] and c.overlap.shape == (2, 2)
    with pytest.raises(ImportError):
        DMRG_EVCont_obj().append_to_rdms("mol")
    cas = CAS_EVCont_obj(4, 2)
    assert (cas.ncas, cas.neleca, cas.cascis, cas.overlap) == (4, 2, [], None)
