"""GPU tests of the active-learning building blocks (evcont_amd/active_learning.py) against the way the
reference computes the same quantities (MD_utils.py:264-299, 363-405): subset energies by slicing the
t-RDM arrays and calling the evaluator again; the farthest-point metric from per-geometry OAO integrals."""
import numpy as np
import pytest
import torch

from evcont_amd.hchain import s_gaussian_mol, hydrogen_chain
from evcont_amd.fci_small import SmallFCI
from evcont_amd.synthetic import make_ao_arrays, make_trdms, pack_rows
from oracle import evcont_oracle as orc
from test_hchain_physics import bundle, bent_chain, chain, train

pytestmark = pytest.mark.gpu


def test_integrals_oao_batch_matches_oracle():
    from evcont_amd.active_learning import integrals_oao_batch
    dev = torch.device("cuda:0")
    for n, B in ((7, 3), (13, 2), (34, 2)):          # odd n, and n > 32 (quarter-step transforms)
        aos = [make_ao_arrays(n, 2, 40 + k, with_ip1=False) for k in range(B)]
        up = lambda name: torch.from_numpy(np.stack([getattr(a, name) for a in aos])).to(dev)
        h1, h2, X = integrals_oao_batch(up("S"), up("hcore"), up("eri"))
        for k, a in enumerate(aos):
            Xo = orc.loewdin_trafo(a.S)
            b = orc.AOBundle(a.S, a.hcore, a.eri, a.ipovlp, a.dhcore, a.eri_ip1, a.aoslices, a.enuc, a.gnuc)
            h1o, h2o = orc.integrals_oao(b, Xo)
            np.testing.assert_allclose(X[k].cpu().numpy(), Xo, rtol=0, atol=1e-11)
            np.testing.assert_allclose(h1[k].cpu().numpy(), h1o, rtol=0, atol=1e-10)
            np.testing.assert_allclose(h2[k].cpu().numpy(), h2o, rtol=0, atol=1e-10)


def test_subspace_energies_batch_vs_scipy():
    import scipy.linalg
    from evcont_amd.active_learning import subspace_energies
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    for T in (1, 2, 5, 8, 19):
        B = 7
        H = rng.standard_normal((B, T, T))
        A = rng.standard_normal((B, T, T))
        S = A @ A.transpose(0, 2, 1) / T + np.eye(T)
        shift = rng.standard_normal(B)
        nroots = min(T, 3)
        e = subspace_energies(torch.from_numpy(H).to(dev), torch.from_numpy(S).to(dev), torch.from_numpy(shift).to(dev),
                              nroots=nroots).cpu().numpy()
        e_shared = subspace_energies(torch.from_numpy(H).to(dev), torch.from_numpy(S[0]).to(dev), None, 1).cpu().numpy()
        for b in range(B):
            ref = scipy.linalg.eigh(H[b], S[b], lower=True, eigvals_only=True)
            np.testing.assert_allclose(e[b], ref[:nroots] + shift[b], rtol=0, atol=1e-11)
            assert abs(e_shared[b, 0] - scipy.linalg.eigh(H[b], S[0], lower=True, eigvals_only=True)[0]) < 1e-11


@pytest.mark.parametrize("lname", ["pack2", "full6"])
def test_subset_energies_match_reevaluation_on_sliced_arrays(lname):
    """Leave-one-out / drop-last energies from ONE contraction per geometry == evaluating again with
    one_rdm[np.ix_(ids, ids)] etc. (what the reference does, MD_utils.py:264-299)."""
    from evcont_amd.active_learning import trajectory_hamiltonians, subset_energies
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAO, ContinuationEvaluator
    dev = torch.device("cuda:0")
    n, T, A, B = 6, 5, 3, 4
    S, one, two = make_trdms(n, T, 21)
    two_l = pack_rows(two, True, True) if lname == "pack2" else two
    aos = [DeviceAO.from_arrays(make_ao_arrays(n, A, 700 + k), dev) for k in range(B)]
    H, E, enuc = trajectory_hamiltonians(DeviceTRDMs(one, two_l, S, dev), aos)
    subsets = [list(range(T - 1))] + [[i for i in range(T) if i != j] for j in range(T)] + [[0, 3], [2]]
    got = subset_energies(H, torch.from_numpy(S).to(dev), enuc, subsets).cpu().numpy()
    for k, ids in enumerate(subsets):
        ix = np.ix_(ids, ids)
        ev = ContinuationEvaluator(DeviceTRDMs(one[ix], two[ix], S[ix], dev), A)
        for b in range(B):
            e_ref = ev.energies(aos[b], 1)[0][0]
            assert abs(got[b, k] - e_ref) < 1e-10, (k, b)
    # and the full set reproduces the energies of the batched call itself
    full = subset_energies(H, torch.from_numpy(S).to(dev), enuc, [list(range(T))]).cpu().numpy()[:, 0]
    np.testing.assert_allclose(full, E.cpu().numpy(), rtol=0, atol=1e-11)


def test_farthest_point_ham_matches_direct_formula():
    from evcont_amd.active_learning import farthest_point_ham
    trn = [hydrogen_chain(4, d, need_grad=False) for d in (1.6, 2.4)]
    traj = [s_gaussian_mol(bent_chain(4, d=1.5 + 0.1 * k, seed=k, amp=0.05), need_grad=False) for k in range(9)]

    def oao(m):
        return orc.integrals_oao(bundle(m), orc.loewdin_trafo(m.S))
    trn_h = [oao(m) for m in trn]
    mins = []
    for m in traj:
        h1, h2 = oao(m)
        mins.append(min(np.sum((h1 - a) ** 2) + 0.5 * np.sum((h2 - b) ** 2) for a, b in trn_h))
    assert farthest_point_ham(traj, trn) == int(np.argmax(mins))


def test_converge_evcont_md_h4(tmp_path):
    """The whole loop on H4: trajectories, selection, growth, convergence files (MD_utils.py:128-502)."""
    from evcont_amd.FCI_EVCont import FCI_EVCont_obj
    from evcont_amd.MD_utils import converge_EVCont_MD
    cont = FCI_EVCont_obj(cisolver=SmallFCI(), cibasis="OAO")
    m0 = s_gaussian_mol(bent_chain(4, d=1.7, seed=2, amp=0.03))
    traj = converge_EVCont_MD(cont, m0, steps=12, dt=5.0, convergence_thresh=1e-4, workdir=str(tmp_path),
                              max_iterations=6)
    assert traj.shape == (12, 4, 3) and np.array_equal(traj[0], m0.coords)
    T = cont.ntrain
    assert 2 <= T <= 7
    n_iter = len(list(tmp_path.glob("en_diff_*.txt")))
    assert n_iter >= 2 and (tmp_path / "overlap.npy").exists() and (tmp_path / "trn_times.txt").exists()
    np.testing.assert_array_equal(np.load(tmp_path / "two_rdm.npy"), cont.two_rdm)
    diffs = [np.loadtxt(tmp_path / f"en_diff_{k}.txt").max() for k in range(n_iter)]
    assert diffs[-1] < diffs[0]                      # adding training points along the trajectory converges it
    if T < 7:                                        # stopped by the criterion, not by max_iterations
        assert diffs[-1] <= 1e-4 and diffs[-2] <= 1e-4
    # the energies written for the last trajectory are those of the final training set
    ens = np.genfromtxt(tmp_path / f"ens_EVCont_{n_iter - 1}.xyz")[:, 1]
    E, _ = orc.energy_with_grad(bundle(m0), cont.one_rdm, cont.two_rdm, cont.overlap)
    assert abs(ens[0] - E) < 1e-8
