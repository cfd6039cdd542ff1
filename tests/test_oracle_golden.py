"""Pin the CPU oracle (oracle/evcont_oracle.py) to golden vectors produced by the
reference's own functions (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import golden_cases, bundle_from_golden
from evcont_amd.synthetic import pack_rows
from oracle import evcont_oracle as orc

CASES = golden_cases()
LAYOUTS = ("full6", "pair5", "elec3", "pack2")


def layout(two, name):
    return {"full6": two, "pair5": pack_rows(two, True, False),
            "elec3": pack_rows(two, False, True), "pack2": pack_rows(two, True, True)}[name]


def test_cases_present():
    assert len(CASES) >= 5


@pytest.mark.parametrize("case", CASES)
def test_loewdin_and_integrals(case, load_golden):
    g = load_golden(case)
    b = bundle_from_golden(g)
    X = orc.loewdin_trafo(b.S)
    np.testing.assert_allclose(X, g["X"], rtol=0, atol=1e-13)
    h1, h2 = orc.integrals_oao(b, X)
    np.testing.assert_allclose(h1, g["h1"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(h2, g["h2"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("case", CASES)
def test_pack_unpack_bitexact(case, load_golden):
    g = load_golden(case)
    n = g["S"].shape[0]
    h2 = g["h2"].copy()
    keep = h2.copy()
    assert np.array_equal(orc.pack_pair_sym(h2, 0.5), g["h2_packed_half"])
    assert np.array_equal(orc.pack_pair_sym(h2, 1.0), g["h2_packed_one"])
    assert np.array_equal(h2, keep)          # out-of-place
    assert np.array_equal(orc.unpack_pair_sym(g["h2_packed_one"], n), g["h2_restored"])


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("lname", LAYOUTS)
@pytest.mark.parametrize("herm", (True, False))
def test_subspace_problem(case, lname, herm, load_golden):
    g = load_golden(case)
    tag = f"{lname}_{'h' if herm else 'nh'}"
    two = layout(g["two_RDM"], lname)
    H = orc.subspace_hamiltonian(g["h1"], g["h2"], g["one_RDM"], two, herm)
    np.testing.assert_allclose(H, g[f"gs_H_{tag}"], rtol=0, atol=1e-12)
    e, c = orc.approximate_ground_state(g["h1"], g["h2"], g["one_RDM"], two, g["S_train"], herm)
    assert abs(e - float(g[f"gs_E_{tag}"])) < 1e-11
    cref = g[f"gs_c_{tag}"]
    assert min(np.abs(c - cref).max(), np.abs(c + cref).max()) < 1e-9
    nroots = len(g[f"ms_E_{tag}"])
    em, cm = orc.approximate_multistate(g["h1"], g["h2"], g["one_RDM"], two, g["S_train"], nroots, herm)
    np.testing.assert_allclose(em, g[f"ms_E_{tag}"], rtol=0, atol=1e-11)
    for k in range(nroots):
        r = g[f"ms_C_{tag}"][k]
        assert min(np.abs(cm[k] - r).max(), np.abs(cm[k] + r).max()) < 1e-8
    b = bundle_from_golden(g)
    et, _ = orc.approximate_ground_state_OAO(b, g["one_RDM"], two, g["S_train"], herm)
    assert abs(et - float(g[f"gsoao_E_{tag}"])) < 1e-11


@pytest.mark.parametrize("case", CASES)
def test_gradient_blocks(case, load_golden):
    g = load_golden(case)
    b = bundle_from_golden(g)
    np.testing.assert_allclose(orc.overlap_grad(b.ipovlp, b.aoslices), g["dS"], rtol=0, atol=0)
    LG = orc.loewdin_trafo_grad_bucketed(b.S)
    np.testing.assert_allclose(LG, g["LG"], rtol=0, atol=2e-12)
    # the tensor is symmetric under (pq)<->(ab): the property the reference relies on
    np.testing.assert_allclose(LG, LG.transpose(2, 3, 0, 1), rtol=0, atol=2e-12)
    dX = orc.derivative_ao_mo_trafo(b)
    np.testing.assert_allclose(dX, g["dX"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(orc.one_el_grad_ao(b), g["h1_jac_ao"], rtol=0, atol=0)
    np.testing.assert_allclose(orc.one_el_grad(b, g["X"], g["dX"]), g["h1_jac"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(orc.one_el_grad(b), g["h1_jac_default"], rtol=0, atol=1e-10)
    sl = [tuple(s) for s in b.aoslices]
    t = orc.two_el_grad(b.eri, g["ewg_G_full6"], g["X"], g["dX"], b.eri_ip1, sl)
    np.testing.assert_allclose(t, g["two_el_grad"], rtol=0, atol=1e-10)
    ge = orc.grad_elec_OAO(b, g["ewg_D_full6"], g["ewg_G_full6"], X=g["X"])
    np.testing.assert_allclose(ge, g["grad_elec"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(orc.grad_elec_OAO(b, g["ewg_D_full6"], g["ewg_G_full6"]),
                               g["grad_elec_default"], rtol=0, atol=1e-10)
    gn = orc.grad_elec_OAO(b, g["nonsym_D"], g["nonsym_G"], X=g["X"])
    np.testing.assert_allclose(gn, g["nonsym_grad_elec"], rtol=0, atol=1e-9)


@pytest.mark.parametrize("case", CASES)
def test_daleckii_krein_matches_reference_pt(case, load_golden):
    """The closed form evaluated on the GPU equals the reference's (degenerate)
    perturbation theory on generic and on exactly degenerate spectra."""
    g = load_golden(case)
    LG = orc.loewdin_trafo_grad_dk(g["S"])
    np.testing.assert_allclose(LG, g["LG"], rtol=0, atol=5e-12)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("lname", LAYOUTS)
def test_energy_with_grad(case, lname, load_golden):
    g = load_golden(case)
    b = bundle_from_golden(g)
    two = layout(g["two_RDM"], lname)
    E, grad, D, G = orc.energy_with_grad(b, g["one_RDM"], two, g["S_train"], True, True)
    assert abs(E - float(g[f"ewg_E_{lname}"])) < 1e-11
    np.testing.assert_allclose(grad, g[f"ewg_grad_{lname}"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(D, g[f"ewg_D_{lname}"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(G, g[f"ewg_G_{lname}"], rtol=0, atol=1e-10)
    # D-K variant (what the device evaluates) stays inside the force budget
    E2, grad2 = orc.energy_with_grad(b, g["one_RDM"], two, g["S_train"], True, False, bucketed=False)
    np.testing.assert_allclose(grad2, g[f"ewg_grad_{lname}"], rtol=0, atol=1e-9)


@pytest.mark.parametrize("case", CASES)
def test_energy_with_grad_nonhermitian(case, load_golden):
    g = load_golden(case)
    b = bundle_from_golden(g)
    E, grad, D, G = orc.energy_with_grad(b, g["one_RDM"], g["two_RDM"], g["S_train"], False, True)
    assert abs(E - float(g["ewg_E_full6_nh"])) < 1e-10
    np.testing.assert_allclose(grad, g["ewg_grad_full6_nh"], rtol=0, atol=1e-8)


def test_grow_and_prune():
    rng = np.random.default_rng(3)
    n = 3
    o = d = t = None
    for T in range(1, 4):
        row = rng.standard_normal(T)
        r1 = [rng.standard_normal((n, n)) for _ in range(T)]
        r2 = [rng.standard_normal((n, n, n, n)) for _ in range(T)]
        o, d, t = orc.grow_trdms(o, d, t, row, r1, r2)
        assert o.shape == (T, T) and d.shape == (T, T, n, n) and t.shape == (T, T) + (n,) * 4
        assert np.array_equal(d[-1, 0], d[0, -1])      # untransposed copy, as in the reference
    o2, d2, t2 = orc.prune_trdms(o, d, t, [0, 2])
    assert o2.shape == (2, 2) and np.array_equal(t2[1, 0], t[2, 0])


def test_oracle_against_reference_large_training_set():
    """T = 40 (the regime of the large-T subspace kernel): the oracle against vectors the REFERENCE produced
    (tests/golden/make_golden_large_T.py: get_energy_with_grad and approximate_multistate on the pack2 layout)."""
    import os
    from conftest import GOLDEN_DIR
    with np.load(os.path.join(GOLDEN_DIR, "largeT_n3t40a2.npz")) as z:
        g = {k: z[k] for k in z.files}
    b = orc.AOBundle(S=g["S"], hcore=g["hcore"], eri=g["eri"], ipovlp=g["ipovlp"], dhcore=g["dhcore"],
                     eri_ip1=g["eri_ip1"], aoslices=g["aoslices"], enuc=float(g["enuc"]), gnuc=g["gnuc"])
    E, grad, D, G = orc.energy_with_grad(b, g["one_RDM"], g["two_RDM_pack2"], g["S_train"], True, True)
    assert abs(E - float(g["ewg_E_pack2"])) < 1e-11
    np.testing.assert_allclose(grad, g["ewg_grad_pack2"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(D, g["ewg_D_pack2"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(np.asarray(G).reshape(g["ewg_G_pack2"].shape), g["ewg_G_pack2"], rtol=0, atol=1e-11)
    X = orc.loewdin_trafo(b.S)
    h1, h2 = orc.integrals_oao(b, X)
    em, cm = orc.approximate_multistate(h1, h2, g["one_RDM"], g["two_RDM_pack2"], g["S_train"], nroots=6)
    np.testing.assert_allclose(em, g["ms_E_pack2"], rtol=0, atol=1e-11)
