"""GPU tests of the drop-in Python API (same names/arguments as the reference's modules),
driven with array-level molecules (no PySCF) against the reference's golden outputs."""
import numpy as np
import pytest
import torch

from conftest import golden_cases, ao_from_golden
from evcont_amd.synthetic import pack_rows

pytestmark = pytest.mark.gpu
CASES = golden_cases()
LAYOUTS = {"full6": (False, False), "pair5": (True, False), "elec3": (False, True), "pack2": (True, True)}


def layout(two, name):
    p, e = LAYOUTS[name]
    return pack_rows(two, p, e) if (p or e) else two


@pytest.fixture(autouse=True)
def _need_gpu():
    assert torch.cuda.is_available()
    from evcont_amd import cache
    cache.clear()


@pytest.mark.parametrize("case", CASES)
def test_electron_integral_utils(case, load_golden):
    import evcont_amd.electron_integral_utils as eiu
    g = load_golden(case)
    mol = ao_from_golden(g)
    n = mol.nao
    X = eiu.get_loewdin_trafo(g["S"])
    np.testing.assert_allclose(X, g["X"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(eiu.get_basis(mol), g["X"], rtol=0, atol=1e-13)
    h1, h2 = eiu.get_integrals(mol, X)
    np.testing.assert_allclose(h1, g["h1"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(h2, g["h2"], rtol=0, atol=1e-12)
    keep = g["h2"].copy()
    assert np.array_equal(eiu.compress_electron_exchange_symmetry(keep, 0.5), g["h2_packed_half"])
    assert np.array_equal(keep, g["h2"])
    assert np.array_equal(eiu.restore_electron_exchange_symmetry(g["h2_packed_one"], n), g["h2_restored"])
    rng = np.random.default_rng(0)
    Tm = rng.standard_normal((n, n))
    a, b = eiu.transform_integrals(g["hcore"], g["eri"], Tm)
    np.testing.assert_allclose(a, np.einsum("ij,ai,bj->ab", g["hcore"], Tm, Tm), rtol=0, atol=1e-11)
    np.testing.assert_allclose(b, np.einsum("ijkl,ai,bj,ck,dl->abcd", g["eri"], Tm, Tm, Tm, Tm, optimize=True),
                               rtol=0, atol=1e-10)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("lname", list(LAYOUTS))
def test_continuation_module(case, lname, load_golden):
    import evcont_amd.ab_initio_eigenvector_continuation as evc
    g = load_golden(case)
    mol = ao_from_golden(g)
    two = layout(g["two_RDM"], lname)
    e, c = evc.approximate_ground_state(g["h1"], g["h2"], g["one_RDM"], two, g["S_train"])
    assert isinstance(e, float) and c.shape == (g["S_train"].shape[0],)
    assert abs(e - float(g[f"gs_E_{lname}_h"])) < 1e-11
    r = g[f"gs_c_{lname}_h"]
    assert min(np.abs(c - r).max(), np.abs(c + r).max()) < 1e-8
    nroots = len(g[f"ms_E_{lname}_h"])
    em, cm = evc.approximate_multistate(g["h1"], g["h2"], g["one_RDM"], two, g["S_train"], nroots=nroots)
    np.testing.assert_allclose(em, g[f"ms_E_{lname}_h"], rtol=0, atol=1e-11)
    assert cm.shape == (nroots, len(c))
    et, ct = evc.approximate_ground_state_OAO(mol, g["one_RDM"], two, g["S_train"])
    assert abs(et - float(g[f"gsoao_E_{lname}_h"])) < 1e-10
    emt, _ = evc.approximate_multistate_OAO(mol, g["one_RDM"], two, g["S_train"], nroots=nroots)
    np.testing.assert_allclose(emt - float(g["enuc"]), g[f"ms_E_{lname}_h"], rtol=0, atol=1e-10)
    # non-Hermitian branch (scipy.linalg.eig, reference :76-81) on the device-assembled H
    e, c = evc.approximate_ground_state(g["h1"], g["h2"], g["one_RDM"], two, g["S_train"], hermitian=False)
    assert abs(e - float(g[f"gs_E_{lname}_nh"])) < 1e-10
    r = g[f"gs_c_{lname}_nh"]
    assert min(np.abs(c - r).max(), np.abs(c + r).max()) < 1e-7
    em, cm = evc.approximate_multistate(g["h1"], g["h2"], g["one_RDM"], two, g["S_train"], nroots=nroots,
                                        hermitian=False)
    np.testing.assert_allclose(em, g[f"ms_E_{lname}_nh"], rtol=0, atol=1e-10)
    assert cm.shape == (nroots, len(c))
    et, _ = evc.approximate_ground_state_OAO(mol, g["one_RDM"], two, g["S_train"], hermitian=False)
    assert abs(et - float(g[f"gsoao_E_{lname}_nh"])) < 1e-10
    emt, _ = evc.approximate_multistate_OAO(mol, g["one_RDM"], two, g["S_train"], nroots=nroots, hermitian=False)
    np.testing.assert_allclose(emt - float(g["enuc"]), g[f"ms_E_{lname}_nh"], rtol=0, atol=1e-10)
    with pytest.raises(AssertionError):
        evc.approximate_ground_state(g["h1"], g["h2"], g["one_RDM"], two.reshape(-1), g["S_train"])


@pytest.mark.parametrize("case", CASES)
def test_gradient_module_blocks(case, load_golden):
    import evcont_amd.ab_initio_gradients_loewdin as gl
    g = load_golden(case)
    mol = ao_from_golden(g)
    np.testing.assert_allclose(gl.get_overlap_grad(mol), g["dS"], rtol=0, atol=0)
    np.testing.assert_allclose(gl.loewdin_trafo_grad(g["S"]), g["LG"], rtol=0, atol=5e-12)
    np.testing.assert_allclose(gl.get_derivative_ao_mo_trafo(mol), g["dX"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(gl.get_one_el_grad_ao(mol), g["h1_jac_ao"], rtol=0, atol=0)
    np.testing.assert_allclose(gl.get_one_el_grad(mol, ao_mo_trafo=g["X"], ao_mo_trafo_grad=g["dX"]),
                               g["h1_jac"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(gl.get_one_el_grad(mol), g["h1_jac_default"], rtol=0, atol=1e-10)
    sl = tuple((int(a), int(b)) for a, b in g["aoslices"])
    t = gl.two_el_grad(g["eri"], g["ewg_G_full6"], g["X"], g["dX"], g["eri_ip1"], sl)
    np.testing.assert_allclose(t, g["two_el_grad"], rtol=0, atol=1e-10)
    D, G = g["ewg_D_full6"], g["ewg_G_full6"]
    np.testing.assert_allclose(gl.get_grad_elec_OAO(mol, D, G, ao_mo_trafo=g["X"]), g["grad_elec"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(gl.get_grad_elec_OAO(mol, D, G), g["grad_elec_default"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(gl.get_grad_elec_OAO(mol, D, G, ao_mo_trafo=g["X"], ao_mo_trafo_grad=g["dX"]),
                               g["grad_elec"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(gl.get_grad_elec_OAO(mol, g["nonsym_D"], g["nonsym_G"], ao_mo_trafo=g["X"]),
                               g["nonsym_grad_elec"], rtol=0, atol=1e-8)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("lname", list(LAYOUTS))
def test_get_energy_with_grad(case, lname, load_golden):
    import evcont_amd.ab_initio_gradients_loewdin as gl
    g = load_golden(case)
    mol = ao_from_golden(g)
    two = layout(g["two_RDM"], lname)
    E, grad = gl.get_energy_with_grad(mol, g["one_RDM"], two, g["S_train"])
    assert isinstance(E, float) and grad.shape == (mol.natm, 3)
    assert abs(E - float(g[f"ewg_E_{lname}"])) < 1e-10
    np.testing.assert_allclose(grad, g[f"ewg_grad_{lname}"], rtol=0, atol=1e-9)
    E2, grad2, D, G = gl.get_energy_with_grad(mol, g["one_RDM"], two, g["S_train"], return_density_matrices=True)
    assert E2 == E
    np.testing.assert_allclose(D, g[f"ewg_D_{lname}"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(G, g[f"ewg_G_{lname}"], rtol=0, atol=1e-10)
    # hermitian=False (reference :341-356 with the eig vector): host eig on the device-assembled H, device gradient
    En, gradn, Dn, Gn = gl.get_energy_with_grad(mol, g["one_RDM"], two, g["S_train"], hermitian=False,
                                                return_density_matrices=True)
    from oracle import evcont_oracle as orc
    from conftest import bundle_from_golden
    Eo, go, Do, Go = orc.energy_with_grad(bundle_from_golden(g), g["one_RDM"], two, g["S_train"], False, True)
    assert abs(En - Eo) < 1e-10
    np.testing.assert_allclose(gradn, go, rtol=0, atol=1e-9)
    np.testing.assert_allclose(Dn, Do, rtol=0, atol=1e-10)
    np.testing.assert_allclose(Gn, np.asarray(Go).reshape(Gn.shape), rtol=0, atol=1e-10)
    if lname == "full6" and "ewg_E_full6_nh" in g:
        # the reference's own output for this branch (tests/golden/make_golden.py)
        assert abs(En - float(g["ewg_E_full6_nh"])) < 1e-10
        np.testing.assert_allclose(gradn, g["ewg_grad_full6_nh"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(Dn, g["ewg_D_full6_nh"], rtol=0, atol=1e-10)
        np.testing.assert_allclose(Gn, g["ewg_G_full6_nh"], rtol=0, atol=1e-10)


def test_sliced_views_and_cache(load_golden):
    """Callers pass non-contiguous slices two_rdm[:j,:j] (evaluate_accuracy_6_31G.py:64-69)."""
    import evcont_amd.ab_initio_gradients_loewdin as gl
    from evcont_amd import cache
    from oracle import evcont_oracle as orc
    from conftest import bundle_from_golden
    g = load_golden("n5t4a2")
    mol = ao_from_golden(g)
    b = bundle_from_golden(g)
    for j in (2, 3, 4):
        one, two, S = g["one_RDM"][:j, :j], g["two_RDM"][:j, :j], g["S_train"][:j, :j]
        assert not two.flags.c_contiguous or j == 4
        E, grad = gl.get_energy_with_grad(mol, one, two, S)
        Eo, go = orc.energy_with_grad(b, one, two, S)
        assert abs(E - Eo) < 1e-10
        np.testing.assert_allclose(grad, go, rtol=0, atol=1e-9)
    # in-place mutation is detected through the content fingerprint
    two = g["two_RDM"].copy()
    E1, _ = gl.get_energy_with_grad(mol, g["one_RDM"], two, g["S_train"])
    two *= 1.01
    E2, _ = gl.get_energy_with_grad(mol, g["one_RDM"], two, g["S_train"])
    Eo, _ = orc.energy_with_grad(b, g["one_RDM"], two, g["S_train"])
    assert abs(E2 - Eo) < 1e-10 and abs(E1 - E2) > 1e-6
    cache.clear()


def test_scanner(load_golden):
    """MD harness surface (MD_utils.py:20-57)."""
    from evcont_amd.MD_utils import get_scanner
    g = load_golden("n6t3a3")
    mol = ao_from_golden(g)
    sc = get_scanner(mol, g["one_RDM"], g["two_RDM"], g["S_train"])
    for attr in ("converged", "ovlp", "one_trdm", "two_trdm", "predicted_one_rdm", "predicted_two_rdm"):
        assert hasattr(sc.base, attr)
    assert sc.base.converged is True and sc.base.predicted_one_rdm is None
    E, grad = sc(mol)
    assert abs(E - float(g["ewg_E_full6"])) < 1e-10
    np.testing.assert_allclose(grad, g["ewg_grad_full6"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(sc.base.predicted_one_rdm, g["ewg_D_full6"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(sc.base.predicted_two_rdm, g["ewg_G_full6"], rtol=0, atol=1e-10)
    assert sc.mol is mol
    sc0 = get_scanner(mol, None, None, None)
    E0, g0 = sc0(mol)
    assert E0 == float(g["enuc"]) and np.array_equal(g0, g["gnuc"])


# (20: integrals beyond the zero-copy limit of the scanner -- uploads, energy-only call + gradient phase; 34: the same
#  through the 64-wide kernels of csrc/pair64.hip)
@pytest.mark.parametrize("n,T,A", [(6, 3, 3), (10, 5, 10), (20, 3, 4), (34, 2, 3)])
def test_default_call_uses_the_compressed_path(n, T, A):
    """The reference's call, unchanged (ab_initio_gradients_loewdin.py:308-379 with the container's 6-index arrays,
    FCI_EVCont.py:106-131; no extra arguments, no environment): with integrals that have the symmetries of real ones
    the default mode keeps the 8-fold compressed copy resident and runs the symmetric pipeline -- same energy and
    forces as the oracle on the ORIGINAL arrays; asking for the predicted RDMs switches to the caller's layout."""
    import evcont_amd.ab_initio_gradients_loewdin as gl
    import evcont_amd.ab_initio_eigenvector_continuation as evc
    from evcont_amd import cache, _lib
    from evcont_amd.synthetic import make_ao_arrays, make_trdms
    from evcont_amd.MD_utils import get_scanner
    from oracle import evcont_oracle as orc
    assert evc.get_trdm_compression() == "auto"
    S, one, two = make_trdms(n, T, 77 + n)
    assert two.ndim == 6
    mol = make_ao_arrays(n, A, 78 + n, ip1_rs_symmetric=True)
    mol.integral_symmetry = None                      # an array-level molecule that does not say: checked numerically
    b = orc.AOBundle(mol.S, mol.hcore, mol.eri, mol.ipovlp, mol.dhcore, mol.eri_ip1, mol.aoslices, mol.enuc, mol.gnuc)
    Eo, go, Do, Go = orc.energy_with_grad(b, one, two, S, True, True)
    E, grad = gl.get_energy_with_grad(mol, one, two, S)
    assert abs(E - Eo) < 1e-10
    np.testing.assert_allclose(grad, go, rtol=0, atol=1e-9)
    ev = evc._evaluator(one, two, S, A, compress="sym8")
    assert ev.t.layout == _lib.LAYOUT_SYM8 and ev._primed, "the default call did not run on the compressed copy"
    assert cache.get(cache.key_of(one, two, S, ("trdms", None))) is None, "the caller's layout was uploaded as well"
    E2, grad2, D, G = gl.get_energy_with_grad(mol, one, two, S, return_density_matrices=True)
    assert abs(E2 - Eo) < 1e-10
    np.testing.assert_allclose(grad2, go, rtol=0, atol=1e-9)
    np.testing.assert_allclose(D, Do, rtol=0, atol=1e-10)
    np.testing.assert_allclose(G, np.asarray(Go).reshape(G.shape), rtol=0, atol=1e-10)   # the un-symmetrised 2-RDM
    # energy-only entry point and the MD scanner take the same decision
    e, c = evc.approximate_ground_state_OAO(mol, one, two, S)
    assert abs(e - Eo) < 1e-10
    sc = get_scanner(mol, one, two, S)
    Es, gs = sc(mol)
    assert sc._hev.t.layout == _lib.LAYOUT_SYM8 and sc._hev.packed
    assert abs(Es - Eo) < 1e-10
    np.testing.assert_allclose(gs, go, rtol=0, atol=1e-9)
    np.testing.assert_allclose(sc.base.predicted_one_rdm, Do, rtol=0, atol=1e-10)
    np.testing.assert_allclose(sc.base.predicted_two_rdm, np.asarray(Go).reshape(G.shape), rtol=0, atol=1e-10)
    # an integral producer that leaves its output packed (s4 / s2kl) in pinned memory, laid out as the scanner's staging
    # slabs are: uploaded from there as it stands (two copies per step when the geometry is copied at all)
    mp = mol.pinned_packed()
    assert mp.eri.shape == (n * (n + 1) // 2,) * 2 and mp.eri_ip1.shape == (3, n, n, n * (n + 1) // 2)
    Ep, gp = sc(mp)
    assert sc._hev.zero_copy or sc._hev._direct_slabs is not None
    assert abs(Ep - Eo) < 1e-10
    np.testing.assert_allclose(gp, go, rtol=0, atol=1e-9)
    Ep2, gp2 = sc(mol)                                  # ... and back to a molecule that is staged by copying
    assert sc._hev._direct_slabs is None and abs(Ep2 - Eo) < 1e-10
    # a general eri_ip1 (what the golden fixtures hold): the same call stays on the caller's layout, and is right
    gen = make_ao_arrays(n, A, 79 + n)
    S2, one2, two2 = make_trdms(n, T, 80 + n)
    bg = orc.AOBundle(gen.S, gen.hcore, gen.eri, gen.ipovlp, gen.dhcore, gen.eri_ip1, gen.aoslices, gen.enuc, gen.gnuc)
    Eg, gg = orc.energy_with_grad(bg, one2, two2, S2)
    E3, grad3 = gl.get_energy_with_grad(gen, one2, two2, S2)
    assert abs(E3 - Eg) < 1e-10
    np.testing.assert_allclose(grad3, gg, rtol=0, atol=1e-9)
    assert cache.get(cache.key_of(one2, two2, S2, ("trdms", "sym8"))) is None
    cache.clear()
