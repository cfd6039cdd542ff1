"""Physical known-answer tests on hydrogen chains in STO-3G, CPU only:

* the closed-form s-Gaussian integrals and ALL their derivative arrays against central finite
  differences (this pins the sign conventions of int1e_ipovlp / hcore_generator / int2e_ip1 that the
  Loewdin gradient formulas assume);
* the small FCI solver (energy functional, H2 known answer);
* the continuation itself, through the oracle: exact at training geometries, variational elsewhere
  (``H6_continuation.py:38-61``, ``evaluate_energetics_training_points.py:63-73``), and the
  analytic nuclear gradient of ``get_energy_with_grad`` against finite differences of the energy.

The same data go through the HIP path in ``tests/test_gpu_hchain.py``.
"""
import numpy as np
import pytest

from evcont_amd.hchain import s_gaussian_mol, hydrogen_chain, boys01
from evcont_amd.fci_small import SmallFCI
from evcont_amd.containers import grow_trdms
from oracle import evcont_oracle as orc


def bundle(m):
    return orc.AOBundle(m.S, m.hcore, m.eri, m.ipovlp, m.dhcore, m.eri_ip1, m.aoslices, m.enuc, m.gnuc)


def bent_chain(n, d=1.8, seed=3, amp=0.25):
    rng = np.random.default_rng(seed)
    R = np.zeros((n, 3))
    R[:, 0] = d * np.arange(n)
    return R + amp * rng.standard_normal((n, 3))


def fd(fun, R, h=1e-5):
    """Central differences of fun(R) w.r.t. every nuclear coordinate -> array (A,3) + fun shape."""
    out = []
    for a in range(R.shape[0]):
        row = []
        for x in range(3):
            Rp, Rm = R.copy(), R.copy()
            Rp[a, x] += h
            Rm[a, x] -= h
            row.append((fun(Rp) - fun(Rm)) / (2 * h))
        out.append(row)
    return np.array(out)


def test_boys():
    t = np.array([0.0, 1e-9, 3e-3, 9.9e-3, 1.01e-2, 0.3, 5.0, 40.0])
    f0, f1 = boys01(t)
    # quadrature reference: F_n(t) = int_0^1 u^(2n) exp(-t u^2) du
    u = np.polynomial.legendre.leggauss(60)
    x, w = 0.5 * (u[0] + 1), 0.5 * u[1]
    for k, tk in enumerate(t):
        assert abs(f0[k] - np.sum(w * np.exp(-tk * x * x))) < 1e-14
        assert abs(f1[k] - np.sum(w * x * x * np.exp(-tk * x * x))) < 1e-14


def test_integral_symmetries_and_h2():
    m = hydrogen_chain(4, 1.7)
    assert np.allclose(np.diag(m.S), 1.0, atol=2e-6)          # STO-3G contraction is normalised
    e = m.eri
    for perm in [(1, 0, 2, 3), (0, 1, 3, 2), (2, 3, 0, 1)]:
        assert np.allclose(e, e.transpose(perm), atol=1e-13)
    # Szabo & Ostlund: S12 = 0.6593 at R = 1.4 a.u.; E_FCI(H2, STO-3G) = -1.1373
    h2 = hydrogen_chain(2, 1.4)
    assert abs(h2.S[0, 1] - 0.6593) < 1e-4
    X = orc.loewdin_trafo(h2.S)
    h1o, h2o = orc.integrals_oao(bundle(h2), X)
    e0, _ = SmallFCI().kernel(h1o, h2o, 2, (1, 1))
    assert abs(e0 + h2.enuc - (-1.13728)) < 2e-5


def test_derivative_integrals_against_finite_differences():
    R = bent_chain(3)
    m = s_gaussian_mol(R)
    n = 3
    # dS/dR_A = -<grad mu|nu>[mu in A] + transpose   (ab_initio_gradients_loewdin.py:13-38)
    dS = orc.overlap_grad(m.ipovlp, m.aoslices)                        # (N,N,A,3)
    dS_fd = fd(lambda r: s_gaussian_mol(r, need_grad=False).S, R)       # (A,3,N,N)
    assert np.allclose(dS, dS_fd.transpose(2, 3, 0, 1), atol=1e-9)
    # hcore_generator: total derivative of hcore
    dh_fd = fd(lambda r: s_gaussian_mol(r, need_grad=False).hcore, R)
    assert np.allclose(m.dhcore, dh_fd, atol=1e-8)
    # int2e_ip1: d(ab|cd)/dR_A = -[a in A](grad a b|cd) - [b in A](grad b a|cd) - [c in A](grad c d|ab) - [d in A](grad d c|ab)
    de_fd = fd(lambda r: s_gaussian_mol(r, need_grad=False).eri, R)    # (A,3,N,N,N,N)
    ip1 = m.eri_ip1
    for A in range(n):
        for x in range(3):
            d = np.zeros((n, n, n, n))
            d[A] -= ip1[x, A]
            d[:, A] -= ip1[x, A]
            d[:, :, A] -= ip1[x, A].transpose(1, 2, 0)
            d[:, :, :, A] -= ip1[x, A].transpose(1, 2, 0)
            assert np.allclose(d, de_fd[A, x], atol=1e-8), (A, x)
    # nuclear repulsion
    g_fd = fd(lambda r: np.float64(s_gaussian_mol(r, need_grad=False).enuc), R)
    assert np.allclose(m.gnuc, g_fd, atol=1e-8)


def test_small_fci_energy_functional_and_transition_rdms():
    m = hydrogen_chain(4, 1.9)
    X = orc.loewdin_trafo(m.S)
    h1, h2 = orc.integrals_oao(bundle(m), X)
    f = SmallFCI()
    es, cs = f.kernel(h1, h2, 4, (2, 2), nroots=3)
    for e, c in zip(es, cs):
        assert abs(f.energy(h1, h2, c, 4, (2, 2)) - e) < 1e-11
    # <a|H|b> from transition RDMs: zero between different eigenstates, E_a on the diagonal
    for a in range(3):
        for b in range(3):
            d1, d2 = f.trans_rdm12(cs[a], cs[b], 4, (2, 2))
            hab = np.sum(h1 * d1) + 0.5 * np.sum(h2 * d2)
            assert abs(hab - (es[a] if a == b else 0.0)) < 1e-10
    d1, d2 = f.make_rdm12(cs[0], 4, (2, 2))
    assert abs(np.trace(d1) - 4) < 1e-12
    assert np.allclose(d2, d2.transpose(2, 3, 0, 1), atol=1e-12)         # electron-pair exchange
    assert abs(np.einsum("ppqq->", d2) - 4 * 3) < 1e-10                 # N(N-1)


def train(geoms, nroots=1):
    """FCI training states (OAO basis) at the given geometries -> (S, one, two, energies)."""
    f = SmallFCI()
    vecs, ens, S, one, two = [], [], None, None, None
    for R in geoms:
        m = s_gaussian_mol(R, need_grad=False)
        n = m.nao
        X = orc.loewdin_trafo(m.S)
        h1, h2 = orc.integrals_oao(bundle(m), X)
        e, c = f.kernel(h1, h2, n, m.nelec)
        vecs.append(c)
        ens.append(e + m.enuc)
        T1 = len(vecs)
        ov = np.array([np.vdot(vecs[-1], v) for v in vecs])
        r1 = np.empty((T1, n, n))
        r2 = np.empty((T1, n, n, n, n))
        for i, v in enumerate(vecs):
            r1[i], r2[i] = f.trans_rdm12(vecs[-1], v, n, m.nelec)
        S, one, two = grow_trdms(S, one, two, ov, r1, r2)
    return S, one, two, np.array(ens)


def chain(n, d):
    R = np.zeros((n, 3))
    R[:, 0] = d * np.arange(n)
    return R


@pytest.fixture(scope="module")
def h6_training():
    spacings = (1.5, 2.0, 2.8)
    return spacings, train([chain(6, d) for d in spacings])


def test_continuation_exact_at_training_points_and_variational(h6_training):
    spacings, (S, one, two, ens) = h6_training
    f = SmallFCI()
    for d, e_fci in zip(spacings, ens):
        m = hydrogen_chain(6, d)
        E, g = orc.energy_with_grad(bundle(m), one, two, S)
        assert abs(E - e_fci) < 1e-9                                   # training point reproduced
    for d in (1.7, 2.3, 3.2):
        m = hydrogen_chain(6, d)
        X = orc.loewdin_trafo(m.S)
        h1, h2 = orc.integrals_oao(bundle(m), X)
        e_fci = f.kernel(h1, h2, 6, (3, 3))[0] + m.enuc
        E, _ = orc.energy_with_grad(bundle(m), one, two, S)
        assert E >= e_fci - 1e-10                                       # variational
        assert E - e_fci < 5e-3                                         # and close (3 training points)


def test_continuation_gradient_against_finite_differences(h6_training):
    _, (S, one, two, _) = h6_training
    R = bent_chain(6, d=1.9, seed=11, amp=0.15)
    E, g = orc.energy_with_grad(bundle(s_gaussian_mol(R)), one, two, S)
    energy = lambda r: np.float64(orc.energy_with_grad(bundle(s_gaussian_mol(r)), one, two, S)[0])
    g_fd = fd(energy, R, h=2e-4)
    assert np.allclose(g, g_fd, atol=2e-7), np.abs(g - g_fd).max()
    assert abs(g.sum(axis=0)).max() < 1e-8                              # translational invariance


def test_h10_fixture_training_points_and_variational(h10_fci):
    """configs[1] with physical inputs (tests/golden/make_h10_fci.py), packed (P,M) layout, CPU oracle."""
    S, one, two = h10_fci["overlap"], h10_fci["one_rdm"], h10_fci["two_rdm_pack2"]
    for d, e_fci in zip(h10_fci["spacings"][[0, 3]], h10_fci["ens"][[0, 3]]):
        E, g = orc.energy_with_grad(bundle(hydrogen_chain(10, float(d))), one, two, S)
        assert abs(E - e_fci) < 1e-8
        assert np.abs(g[:, 1:]).max() < 1e-9                 # linear chain along x: no transverse force
    E, g = orc.energy_with_grad(bundle(s_gaussian_mol(h10_fci["R_test"])), one, two, S)
    assert float(h10_fci["e_fci_test"]) - 1e-9 <= E < float(h10_fci["e_fci_test"]) + 0.15   # 5 equidistant linear training chains vs a bent one
