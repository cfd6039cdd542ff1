"""GPU parity tests: every HIP kernel and the fused pipeline, through the C ABI, against
(a) the golden vectors produced by the reference itself and (b) the CPU oracle on seeded
inputs.  Tolerances: energies 1e-8 Ha, forces 1e-6 Ha/Bohr are the north-star budget; the
tests hold the kernels to 1e-10 / 1e-9 on these O(1..10)-sized problems."""
import numpy as np
import pytest
import torch

from conftest import golden_cases, bundle_from_golden, ao_from_golden
from evcont_amd.synthetic import make_ao_arrays, make_trdms, pack_rows

pytestmark = pytest.mark.gpu

CASES = golden_cases()
LAYOUTS = {"full6": (False, False), "pair5": (True, False), "elec3": (False, True), "pack2": (True, True)}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def up(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).to(dev)


def layout(two, name):
    p, e = LAYOUTS[name]
    return pack_rows(two, p, e) if (p or e) else two


def same_up_to_sign(a, b, tol):
    return min(np.abs(a - b).max(), np.abs(a + b).max()) < tol


# ------------------------------------------------------------------ streaming GEMVs
@pytest.mark.parametrize("rows,cols", [(1, 1), (3, 7), (8, 512), (9, 513), (15, 5050), (55, 14365),
                                       (210, 40000), (400, 900), (21, 131072 + 3)])
def test_gemv_rows_cols(rows, cols, dev):
    from evcont_amd import ops
    rng = np.random.default_rng(rows * 1000 + cols)
    A = rng.standard_normal((rows, cols))
    v = rng.standard_normal(cols)
    w = rng.standard_normal(rows)
    Ad = ops.padded_matrix(up(A, dev))
    y = ops.gemv_rows(Ad, cols, up(v, dev), alpha=0.5).cpu().numpy()
    ref = 0.5 * (A @ v)
    assert np.abs(y - ref).max() <= 1e-12 * max(1.0, np.abs(A).sum(1).max())
    o = ops.gemv_cols(Ad, cols, up(w, dev)).cpu().numpy()
    refc = w @ A
    assert np.abs(o - refc).max() <= 1e-12 * max(1.0, np.abs(A).sum(0).max())
    # determinism: two launches give bit-identical sums
    y2 = ops.gemv_rows(Ad, cols, up(v, dev), alpha=0.5).cpu().numpy()
    assert np.array_equal(y, y2)


def test_gemv_rows_linearity_large(dev):
    """Size-independent property at the H30/T=20 packed row length (405450 columns)."""
    from evcont_amd import ops
    rows, cols = 16, 405450
    g = torch.Generator(device=dev).manual_seed(5)
    A = torch.randn((rows, cols), generator=g, device=dev, dtype=torch.float64)
    v1 = torch.randn(cols, generator=g, device=dev, dtype=torch.float64)
    v2 = torch.randn(cols, generator=g, device=dev, dtype=torch.float64)
    Ad = ops.padded_matrix(A)
    y1, y2 = ops.gemv_rows(Ad, cols, v1), ops.gemv_rows(Ad, cols, v2)
    y12 = ops.gemv_rows(Ad, cols, (v1 + 2.0 * v2).contiguous())
    assert (y12 - (y1 + 2.0 * y2)).abs().max().item() < 1e-9
    ref = A @ v1
    assert (y1 - ref).abs().max().item() < 1e-9
    w = torch.randn(rows, generator=g, device=dev, dtype=torch.float64)
    o = ops.gemv_cols(Ad, cols, w)
    assert (o - w @ A).abs().max().item() < 1e-11


# ------------------------------------------------------------------ codecs
@pytest.mark.parametrize("case", CASES)
def test_pack_unpack_bitexact(case, load_golden, dev):
    from evcont_amd import ops
    g = load_golden(case)
    n = g["S"].shape[0]
    h2 = up(g["h2"], dev)
    assert np.array_equal(ops.pack_pair_sym(h2, 0.5).cpu().numpy(), g["h2_packed_half"])
    assert np.array_equal(ops.pack_pair_sym(h2, 1.0).cpu().numpy(), g["h2_packed_one"])
    assert np.array_equal(h2.cpu().numpy(), g["h2"])                     # input untouched
    padded = ops.pack_pair_sym(h2, 1.0, pad_to=len(g["h2_packed_one"]) + 37).cpu().numpy()
    assert np.all(padded[len(g["h2_packed_one"]):] == 0.0)
    r = ops.unpack_pair_sym(up(g["h2_packed_one"], dev), n).cpu().numpy()
    assert np.array_equal(r, g["h2_restored"])


@pytest.mark.parametrize("n", [1, 2, 7, 13, 30])
def test_pack_roundtrip_sizes(n, dev):
    from evcont_amd import ops
    rng = np.random.default_rng(n)
    a = rng.standard_normal((n * n, n * n))
    a = (a + a.T).reshape(n, n, n, n)
    p = ops.pack_pair_sym(up(a, dev), 1.0)
    assert p.numel() == n * n * (n * n + 1) // 2
    assert np.array_equal(ops.unpack_pair_sym(p, n).cpu().numpy(), a)


# ------------------------------------------------------------------ Loewdin + transforms
@pytest.mark.parametrize("case", CASES)
def test_loewdin_and_integrals_golden(case, load_golden, dev):
    from evcont_amd import ops
    g = load_golden(case)
    X, U, s, h1 = ops.loewdin(up(g["S"], dev), up(g["hcore"], dev))
    np.testing.assert_allclose(X.cpu().numpy(), g["X"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(h1.cpu().numpy(), g["h1"], rtol=0, atol=1e-12)
    Un, sn = U.cpu().numpy(), s.cpu().numpy()
    np.testing.assert_allclose((Un * sn) @ Un.T, g["S"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(Un.T @ Un, np.eye(len(sn)), rtol=0, atol=1e-13)
    h2, k3 = ops.four_index_transform(up(g["eri"], dev), X, want_three_quarter=True)
    np.testing.assert_allclose(h2.cpu().numpy(), g["h2"], rtol=0, atol=1e-12)
    Xn = g["X"]
    K = np.einsum("abcd,bj,ck,dl->jkla", g["eri"], Xn, Xn, Xn, optimize=True)
    np.testing.assert_allclose(k3.cpu().numpy(), K, rtol=0, atol=1e-12)


@pytest.mark.parametrize("n", [1, 3, 10, 13, 16, 17, 30, 33])
def test_quarter_transform_asymmetric(n, dev):
    """Asymmetric operands catch a transposed fragment map (guide §3)."""
    from evcont_amd import ops
    rng = np.random.default_rng(100 + n)
    t = rng.standard_normal((n, n, n, n))
    Cm = rng.standard_normal((n, n))
    o = ops.quarter_transform(up(t, dev), up(Cm, dev), False).cpu().numpy()
    np.testing.assert_allclose(o, np.einsum("abcd,dq->qabc", t, Cm), rtol=0, atol=1e-12 * n)
    o = ops.quarter_transform(up(t, dev), up(Cm, dev), True).cpu().numpy()
    np.testing.assert_allclose(o, np.einsum("abcd,qd->qabc", t, Cm), rtol=0, atol=1e-12 * n)
    full = ops.four_index_transform(up(t, dev), up(Cm, dev), True).cpu().numpy()
    ref = np.einsum("ijkl,ai,bj,ck,dl->abcd", t, Cm, Cm, Cm, Cm, optimize=True)
    np.testing.assert_allclose(full, ref, rtol=0, atol=1e-11 * n * n)


@pytest.mark.parametrize("n", [2, 5, 20, 30, 31, 58])
def test_loewdin_sizes(n, dev):
    from evcont_amd import ops
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, n))
    S = B @ B.T / n + np.eye(n)
    X, U, s = ops.loewdin(up(S, dev))
    Xn = X.cpu().numpy()
    np.testing.assert_allclose(Xn @ S @ Xn, np.eye(n), rtol=0, atol=1e-12)
    w = np.linalg.eigvalsh(S)
    np.testing.assert_allclose(np.sort(s.cpu().numpy()), w, rtol=0, atol=1e-13 * n)


# ------------------------------------------------------------------ subspace problem
@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("lname", list(LAYOUTS))
def test_subspace_solve_golden(case, lname, load_golden, dev):
    from evcont_amd import ops
    from evcont_amd.evaluator import layout_shape
    g = load_golden(case)
    T, n = g["S_train"].shape[0], g["S"].shape[0]
    two = layout(g["two_RDM"], lname)
    nd = two.ndim
    rows, cols = layout_shape(nd, T, n)
    A = ops.padded_matrix(up(two.reshape(rows, cols), dev))
    h2 = up(g["h2"], dev)
    if nd in (3, 2):
        v, alpha = ops.pack_pair_sym(h2, 0.5, pad_to=A.shape[1]), 1.0
    else:
        v, alpha = h2.reshape(-1), 0.5
    h2rows = ops.gemv_rows(A, cols, v, alpha)
    D = ops.padded_matrix(up(g["one_RDM"].reshape(T * T, n * n), dev))
    h1rows = ops.gemv_rows(D, n * n, up(g["h1"], dev).reshape(-1))
    nroots = len(g[f"ms_E_{lname}_h"])
    ev, vec, w2, w1, H = ops.subspace_solve(h1rows, h2rows, up(g["S_train"], dev), nd, nroots, 0.25)
    Href = g[f"gs_H_{lname}_h"]
    lo = np.tril_indices(T)
    np.testing.assert_allclose(H.cpu().numpy()[lo], Href[lo], rtol=0, atol=1e-12)
    np.testing.assert_allclose(H.cpu().numpy(), Href, rtol=0, atol=1e-12)   # incl. the reference's upper-triangle quirk
    np.testing.assert_allclose(ev.cpu().numpy() - 0.25, g[f"ms_E_{lname}_h"], rtol=0, atol=1e-11)
    assert abs(ev[0].item() - 0.25 - float(g[f"gs_E_{lname}_h"])) < 1e-11
    vn = vec.cpu().numpy()
    for k in range(nroots):
        assert same_up_to_sign(vn[k], g[f"ms_C_{lname}_h"][k], 1e-8)
    # S-orthonormal rows (evcont.py:169-173)
    np.testing.assert_allclose(vn @ g["S_train"] @ vn.T, np.eye(nroots), rtol=0, atol=1e-11)
    c = vn[0]
    np.testing.assert_allclose(w1.cpu().numpy(), np.outer(c, c).ravel(), rtol=0, atol=1e-14)
    if nd in (5, 2):
        m = 2 * np.outer(c, c)
        m[np.diag_indices(T)] *= 0.5
        np.testing.assert_allclose(w2.cpu().numpy(), m[lo], rtol=0, atol=1e-14)
    else:
        np.testing.assert_allclose(w2.cpu().numpy(), np.outer(c, c).ravel(), rtol=0, atol=1e-14)


# ------------------------------------------------------------------ fused pipeline vs the reference's outputs
@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("lname", list(LAYOUTS))
def test_energy_with_grad_golden(case, lname, load_golden, dev):
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAO, ContinuationEvaluator
    g = load_golden(case)
    ao = DeviceAO.from_arrays(ao_from_golden(g), dev)
    t = DeviceTRDMs(g["one_RDM"], layout(g["two_RDM"], lname), g["S_train"], dev)
    ev = ContinuationEvaluator(t, ao.natm)
    E, grad, D, G = ev.energy_with_grad(ao, return_density_matrices=True)
    assert abs(E - float(g[f"ewg_E_{lname}"])) < 1e-10          # budget: 1e-8 Ha
    np.testing.assert_allclose(grad, g[f"ewg_grad_{lname}"], rtol=0, atol=1e-9)   # budget: 1e-6 Ha/Bohr
    np.testing.assert_allclose(D, g[f"ewg_D_{lname}"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(G, g[f"ewg_G_{lname}"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(np.tril(ev.hmat.cpu().numpy()), np.tril(g[f"gs_H_{lname}_h"]), rtol=0, atol=1e-12)
    # energy-only / multistate entry (approximate_multistate_OAO)
    nroots = len(g[f"ms_E_{lname}_h"])
    es, cs = ev.energies(ao, nroots)
    np.testing.assert_allclose(es - float(g["enuc"]), g[f"ms_E_{lname}_h"], rtol=0, atol=1e-10)
    assert abs(es[0] - float(g[f"gsoao_E_{lname}_h"])) < 1e-10
    # repeatability: a second evaluation on the same buffers is bit-identical
    E2, grad2 = ev.energy_with_grad(ao)
    assert E2 == E and np.array_equal(grad, grad2)


@pytest.mark.parametrize("case", CASES)
def test_grad_elec_given_rdms_golden(case, load_golden, dev):
    from evcont_amd.evaluator import DeviceAO
    from evcont_amd.gradients import grad_elec_oao_device
    g = load_golden(case)
    ao = DeviceAO.from_arrays(ao_from_golden(g), dev)
    ge = grad_elec_oao_device(ao, up(g["ewg_D_full6"], dev), up(g["ewg_G_full6"], dev)).cpu().numpy()
    np.testing.assert_allclose(ge, g["grad_elec"], rtol=0, atol=1e-9)
    gn = grad_elec_oao_device(ao, up(g["nonsym_D"], dev), up(g["nonsym_G"], dev)).cpu().numpy()
    np.testing.assert_allclose(gn, g["nonsym_grad_elec"], rtol=0, atol=1e-8)


# ------------------------------------------------------------------ vs the oracle on BASELINE-shaped inputs
@pytest.mark.parametrize("n,T,A,sizes,lname", [
    (10, 5, 10, None, "full6"),            # config 2: H10, 5 training states
    (10, 5, 10, None, "pack2"),
    (13, 4, 3, (9, 2, 2), "pack2"),        # config 4 shape: H2O 6-31G (odd M)
    (13, 4, 3, (9, 2, 2), "pair5"),
    (16, 3, 4, None, "elec3"),
])
def test_energy_with_grad_vs_oracle(n, T, A, sizes, lname, dev):
    from oracle import evcont_oracle as orc
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAO, ContinuationEvaluator
    ao = make_ao_arrays(n, A, 77 + n, ao_sizes=sizes)
    S, one, two = make_trdms(n, T, 99 + n)
    two_l = layout(two, lname)
    b = orc.AOBundle(**{k: getattr(ao, k) for k in ("S", "hcore", "eri", "ipovlp", "dhcore", "eri_ip1",
                                                     "aoslices", "enuc", "gnuc")})
    Eo, go, Do, Go = orc.energy_with_grad(b, one, two_l, S, True, True)
    dao = DeviceAO.from_arrays(ao, dev)
    ev = ContinuationEvaluator(DeviceTRDMs(one, two_l, S, dev), dao.natm)
    E, grad, D, G = ev.energy_with_grad(dao, True)
    assert abs(E - Eo) < 1e-9
    np.testing.assert_allclose(grad, go, rtol=0, atol=1e-8)
    np.testing.assert_allclose(D, Do, rtol=0, atol=1e-10)
    np.testing.assert_allclose(G, Go, rtol=0, atol=1e-10)


def test_h30_layouts_agree(dev):
    """BASELINE config 3 at full N (N=30, A=30) with T reduced so that the 6-index array fits the
    test budget: the four layouts must give the same energy and forces (size-independent property
    verified on the reference in SURVEY.md §4)."""
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAO, ContinuationEvaluator
    n, T, A = 30, 4, 30
    ao = make_ao_arrays(n, A, 1234 + 3)
    S, one, two = make_trdms(n, T, 4321)
    dao = DeviceAO.from_arrays(ao, dev)
    res = {}
    for lname in LAYOUTS:
        ev = ContinuationEvaluator(DeviceTRDMs(one, layout(two, lname), S, dev), A)
        res[lname] = ev.energy_with_grad(dao)
    E0, g0 = res["full6"]
    for lname, (E, g) in res.items():
        assert abs(E - E0) < 1e-10, lname
        np.testing.assert_allclose(g, g0, rtol=0, atol=1e-9, err_msg=lname)


@pytest.mark.parametrize("lname", ["pack2", "full6"])
@pytest.mark.parametrize("world", [2, 3])
def test_phase_api_emulated_pair_sharding(lname, world, load_golden, dev):
    """The three-phase C entry points on row slices (what each rank of the pair-sharded multi-GPU run
    executes), with the two collectives emulated on one device: concatenate the H rows, sum the
    partial gradients.  Must reproduce the reference's single-process result."""
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAO, ContinuationEvaluator, layout_shape
    from evcont_amd.distributed import shard_rows
    g = load_golden("n5t4a2")
    ao = DeviceAO.from_arrays(ao_from_golden(g), dev)
    two = layout(g["two_RDM"], lname)
    T, n = 4, 5
    rows, _ = layout_shape(two.ndim, T, n)
    evs = []
    for r in range(world):
        r0, r1 = shard_rows(rows, world, r)
        evs.append(ContinuationEvaluator(DeviceTRDMs(g["one_RDM"], two, g["S_train"], dev, row_range=(r0, r1)), ao.natm))
    parts = [ev.phase_hamiltonian(ao).clone() for ev in evs]
    rows_all = torch.cat(parts).contiguous()
    assert rows_all.numel() == rows
    total = torch.zeros_like(evs[0].grad)
    for r, ev in enumerate(evs):
        ev.phase_solve(ao, rows_all, 1)
        ev.phase_gradient(ao, partial_rank=(r != 0))
        total += ev.grad
    torch.cuda.synchronize()
    for ev in evs:
        assert abs(ev.energy[0].item() - float(g[f"ewg_E_{lname}"])) < 1e-10
    np.testing.assert_allclose(total.cpu().numpy(), g[f"ewg_grad_{lname}"], rtol=0, atol=1e-9)
    # the predicted RDMs of the pair-sharded mode (distributed.PairShardedContinuation(return_density_matrices=True)):
    # every rank holds the complete 1-RDM, the 2-RDM is the SUM of the ranks' unpacked partial ones
    Gsum = torch.zeros_like(evs[0].g_pred)
    for ev in evs:
        np.testing.assert_allclose(ev.d_pred.cpu().numpy(), g[f"ewg_D_{lname}"], rtol=0, atol=1e-10)
        Gsum += ev.g_pred
    np.testing.assert_allclose(Gsum.cpu().numpy(), g[f"ewg_G_{lname}"], rtol=0, atol=1e-10)
