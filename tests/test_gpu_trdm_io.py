"""GPU tests of the on-disk formats (evcont_amd/trdm_io.py): the per-pair directories of the Zundel
scripts, the six-index / packed checkpoints, and the prefix (sub-basis) rule."""
import numpy as np
import pytest
import torch

from evcont_amd.synthetic import make_ao_arrays, make_trdms, pack_rows

pytestmark = pytest.mark.gpu


def energies(trd, aos, A):
    from evcont_amd.evaluator import ContinuationEvaluator
    ev = ContinuationEvaluator(trd, A)
    return [ev.energy_with_grad(a) for a in aos]


def test_pair_directories_checkpoints_and_prefix(tmp_path):
    from evcont_amd import trdm_io
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAO
    dev = torch.device("cuda:0")
    n, T, A = 5, 4, 2                                   # odd N*N: ragged packed rows
    S, one, two = make_trdms(n, T, 8)
    # the per-pair files carry only a >= b: make the upper one-body blocks the untransposed lower ones,
    # which is what every container of the reference produces and what the Zundel loader reconstructs
    ia, ib = np.tril_indices(T)
    one[ib, ia] = one[ia, ib]
    two[ib, ia] = two[ia, ib]
    packed = pack_rows(two, True, True)
    aos = [DeviceAO.from_arrays(make_ao_arrays(n, A, 60 + k), dev) for k in range(2)]
    ref = energies(DeviceTRDMs(one, packed, S, dev), aos, A)

    trdm_io.save_pair_directories(str(tmp_path / "pairs"), S, one, packed)
    assert (tmp_path / "pairs" / "MPS_cross_3_1" / "two_rdm.npy").exists()
    assert np.load(tmp_path / "pairs" / "MPS_cross_2_0" / "two_rdm.npy").shape == (packed.shape[1],)
    t1 = trdm_io.load_pair_directories(str(tmp_path / "pairs"), T, dev)
    assert t1.layout == 2 and t1.T == T and torch.equal(t1.two[:, : t1.cols].cpu(), torch.from_numpy(packed))

    trdm_io.save_pair_directories(str(tmp_path / "pairs6"), S, one, two)          # from the six-index array
    t2 = trdm_io.load_pair_directories(str(tmp_path / "pairs6"), T, dev)

    np.save(tmp_path / "overlap.npy", S)
    np.save(tmp_path / "one_rdm.npy", one)
    np.save(tmp_path / "two_rdm.npy", two)                                         # six-index checkpoint
    t3 = trdm_io.load_checkpoint(str(tmp_path), dev)
    np.save(tmp_path / "overlap_7.npy", S)
    np.save(tmp_path / "one_rdm_7.npy", one)
    np.save(tmp_path / "two_rdm_7.npy", packed)                                    # packed checkpoint, suffixed
    t4 = trdm_io.load_checkpoint(str(tmp_path), dev, suffix="_7")
    for t in (t1, t2, t3, t4):
        assert torch.equal(t.two[:, : t.cols], t1.two[:, : t1.cols])               # bit-identical rows
        for (E, g), (E0, g0) in zip(energies(t, aos, A), ref):
            assert abs(E - E0) < 1e-12 and np.abs(g - g0).max() < 1e-12

    # sub-basis rule (05_Zundel_test_potential_energy.py:114-131): first k states = first k(k+1)/2 rows
    for k in (1, 2, 3):
        sub = trdm_io.prefix(t1, k)
        assert sub.two.data_ptr() == t1.two.data_ptr() and sub.rows_total == k * (k + 1) // 2
        ix = np.ix_(range(k), range(k))
        want = energies(DeviceTRDMs(one[ix], two[ix], S[ix], dev), aos, A)
        for (E, g), (E0, g0) in zip(energies(sub, aos, A), want):
            assert abs(E - E0) < 1e-10 and np.abs(g - g0).max() < 1e-9


@pytest.mark.parametrize("compress", [None, "sym8"])
def test_learning_curve_from_one_contraction(compress):
    """Energy vs. number of training states (the scan of 05_Zundel_test_potential_energy.py) from ONE H(R):
    prefix subsets through subset_energies == evaluating each prefix training set (also on the 8-fold
    compressed resident layout, whose pair rows obey the same prefix rule)."""
    from evcont_amd import trdm_io
    from evcont_amd.active_learning import trajectory_hamiltonians, subset_energies
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAO, ContinuationEvaluator
    dev = torch.device("cuda:0")
    n, T, A = 6, 5, 3
    S, one, two = make_trdms(n, T, 33)
    trd = DeviceTRDMs(one, pack_rows(two, True, True), S, dev, compress=compress)
    aos = [DeviceAO.from_arrays(make_ao_arrays(n, A, 90 + k), dev) for k in range(3)]
    H, E, enuc = trajectory_hamiltonians(trd, aos)
    curve = subset_energies(H, trd.S, enuc, [list(range(k)) for k in range(1, T + 1)]).cpu().numpy()
    for k in range(1, T + 1):
        ev = ContinuationEvaluator(trdm_io.prefix(trd, k), A)
        for b, ao in enumerate(aos):
            assert abs(curve[b, k - 1] - ev.energies(ao, 1)[0][0]) < 1e-10
    assert np.all(np.diff(curve, axis=1) <= 1e-12)       # a larger subspace never raises the energy
