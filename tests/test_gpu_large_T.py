"""Training sets with more than 64 states (the reference has no bound on T: its Zundel learning curve evaluates 80 and
100 training states, scripts/MD/Zundel_thermodynamics/continuation/05_Zundel_test_potential_energy.py:182-210, sliced
by the rule of :114-131; converge_EVCont_MD grows T without bound, MD_utils.py:128-502).

Every case is held to ``oracle.energy_with_grad`` / ``oracle.approximate_ground_state`` on the ORIGINAL rows
(get_energy_with_grad, ab_initio_gradients_loewdin.py:308-379; eigh(H, S), ab_initio_eigenvector_continuation.py:73-88).
Tolerances: |dE| <= 1e-8 Ha, |dgrad| <= 1e-6 Ha/Bohr (BASELINE.json north_star); asserted two orders tighter."""
import numpy as np
import pytest
import torch

from evcont_amd.synthetic import make_ao_arrays, make_trdms, pack_rows
from oracle import evcont_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _bundle(ao):
    return orc.AOBundle(ao.S, ao.hcore, ao.eri, ao.ipovlp, ao.dhcore, ao.eri_ip1, ao.aoslices, ao.enuc, ao.gnuc)


def _bundle_from_device(ao):
    c = lambda t: t.cpu().numpy()
    return orc.AOBundle(S=c(ao.S), hcore=c(ao.hcore), eri=c(ao.eri), ipovlp=c(ao.ipovlp), dhcore=c(ao.dhcore),
                        eri_ip1=c(ao.eri_ip1), aoslices=c(ao.aoslices), enuc=ao.enuc, gnuc=c(ao.gnuc))


@pytest.mark.parametrize("T", [33, 47, 64, 65, 80, 100, 128, 130])
def test_small_n_every_layout_single_geometry(T):
    """N = 4 (cheap oracle), all four reference layouts + sym8, one geometry per call (MD regime): the large-T subspace
    kernel (LDS-resident up to 128, global beyond) inside the fused pipeline."""
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAO, ContinuationEvaluator
    n, A = 4, 2
    S, one, two = make_trdms(n, T, 3000 + T)
    ao = make_ao_arrays(n, A, 3100 + T, ip1_rs_symmetric=True)
    two_p = pack_rows(two, True, True)
    Eo, go = orc.energy_with_grad(_bundle(ao), one, two_p, S)
    layouts = {"full6": two, "pair5": pack_rows(two, True, False), "elec3": pack_rows(two, False, True), "pack2": two_p}
    if T > 100:
        layouts = {"pack2": two_p, "full6": two}
    for name, arr in layouts.items():
        ev = ContinuationEvaluator(DeviceTRDMs(one, arr, S, DEV), A)
        E, g = ev.energy_with_grad(DeviceAO.from_arrays(ao, DEV))
        assert abs(E - Eo) < 1e-10 and np.abs(g - go).max() < 1e-8, (name, T, abs(E - Eo), np.abs(g - go).max())
    ev = ContinuationEvaluator(DeviceTRDMs(one, two_p, S, DEV, compress="sym8"), A)
    E, g = ev.energy_with_grad(DeviceAO.from_arrays(ao, DEV, pack_ip1=True, pack_eri=True))
    assert abs(E - Eo) < 1e-10 and np.abs(g - go).max() < 1e-8, ("sym8", T)


@pytest.mark.parametrize("T,G", [(40, 32), (80, 32), (100, 17), (100, 5), (128, 13)])
def test_small_n_batched(T, G):
    """Batches (K5/K8 matrix-core kernels with hundreds of row tiles, the transposed weight copy for 5050+ rows)."""
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAOBatch, BatchedEvaluator
    n, A = 6, 3
    S, one, two = make_trdms(n, T, 3300 + T)
    two_p = pack_rows(two, True, True)
    del two
    aos = [make_ao_arrays(n, A, 3400 + 50 * T + k, ip1_rs_symmetric=True) for k in range(G)]
    slots = sorted({0, G // 2, G - 1})
    want = {k: orc.energy_with_grad(_bundle(aos[k]), one, two_p, S) for k in slots}
    for comp in ("sym8", None):
        be = BatchedEvaluator(DeviceTRDMs(one, two_p, S, DEV, compress=comp), A, G)
        packed = comp is not None
        E, grad = be.energies_with_grads(DeviceAOBatch.from_arrays(aos, DEV, pack_ip1=packed, pack_eri=packed))
        for k in slots:
            assert abs(E[k] - want[k][0]) < 1e-10, (comp, k, abs(E[k] - want[k][0]))
            np.testing.assert_allclose(grad[k], want[k][1], rtol=0, atol=1e-8)
        # a second call finds the cached factorisation of S_train: identical results
        E2, grad2 = be.energies_with_grads(DeviceAOBatch.from_arrays(aos, DEV, pack_ip1=packed, pack_eri=packed))
        assert np.array_equal(E, E2) and np.array_equal(grad, grad2)


def test_multistate_and_learning_curve_T100():
    """approximate_multistate at T = 100 (nroots = 4) and the sub-basis learning curve k = 20 ... 100 of
    05_Zundel_test_potential_energy.py:114-131 (the pairs of the first k states are the first k(k+1)/2 rows) against the
    oracle on the sliced arrays."""
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAO, ContinuationEvaluator
    from evcont_amd import trdm_io
    n, A, T = 5, 2, 100
    S, one, two = make_trdms(n, T, 3501)
    two_p = pack_rows(two, True, True)
    del two
    ao = make_ao_arrays(n, A, 3502, ip1_rs_symmetric=True)
    full = DeviceTRDMs(one, two_p, S, DEV)
    e, c = ContinuationEvaluator(full, A).energies(DeviceAO.from_arrays(ao, DEV, energy_only=True), nroots=4)
    b = _bundle(ao)
    X = orc.loewdin_trafo(b.S)
    h1, h2 = orc.integrals_oao(b, X)
    eo, co = orc.approximate_multistate(h1, h2, one, two_p, S, nroots=4)
    np.testing.assert_allclose(e, eo + b.enuc, rtol=0, atol=1e-10)
    for k in range(4):
        assert min(np.abs(c[k] - co[k]).max(), np.abs(c[k] + co[k]).max()) < 1e-8
    for k in (20, 33, 50, 64, 65, 80, 100):
        sub = trdm_io.prefix(full, k)
        E, g = ContinuationEvaluator(sub, A).energy_with_grad(DeviceAO.from_arrays(ao, DEV))
        P = k * (k + 1) // 2
        Eo, go = orc.energy_with_grad(b, one[:k, :k], two_p[:P], S[:k, :k])
        assert abs(E - Eo) < 1e-10 and np.abs(g - go).max() < 1e-8, (k, abs(E - Eo))


def test_nonhermitian_T80():
    """hermitian=False at T = 80 (the eig branch, ab_initio_eigenvector_continuation.py:76-88 and
    ab_initio_gradients_loewdin.py:341-356): device-assembled H, host scipy.linalg.eig as in the reference."""
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAO, ContinuationEvaluator
    n, A, T = 4, 2, 80
    S, one, two = make_trdms(n, T, 3601)
    ao = make_ao_arrays(n, A, 3602)
    for arr in (two, pack_rows(two, True, True)):
        ev = ContinuationEvaluator(DeviceTRDMs(one, arr, S, DEV), A)
        E, g = ev.energy_with_grad_nonhermitian(DeviceAO.from_arrays(ao, DEV))
        Eo, go = orc.energy_with_grad(_bundle(ao), one, arr, S, hermitian=False)
        assert abs(E - Eo) < 1e-9 and np.abs(g - go).max() < 1e-7, (arr.ndim, abs(E - Eo), np.abs(g - go).max())


def test_warm_start_T100():
    """EVC_FLAG_WARM_START with the large-T solver: a slowly varying sequence of geometries, every step against the cold
    evaluation."""
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAO, ContinuationEvaluator
    n, A, T = 5, 2, 100
    S, one, two = make_trdms(n, T, 3701)
    two_p = pack_rows(two, True, True)
    del two
    trd = DeviceTRDMs(one, two_p, S, DEV)
    warm, cold = ContinuationEvaluator(trd, A, warm_start=True), ContinuationEvaluator(trd, A)
    a0, a1 = make_ao_arrays(n, A, 3702), make_ao_arrays(n, A, 3703)
    for step in range(4):
        lam = 0.01 * step
        ao = make_ao_arrays(n, A, 3702)
        for f in ("S", "hcore", "eri", "ipovlp", "dhcore", "eri_ip1", "gnuc"):
            setattr(ao, f, (1 - lam) * getattr(a0, f) + lam * getattr(a1, f))
        d = DeviceAO.from_arrays(ao, DEV)
        Ew, gw = warm.energy_with_grad(d)
        Ec, gc = cold.energy_with_grad(d)
        assert abs(Ew - Ec) < 1e-11 and np.abs(gw - gc).max() < 1e-10, (step, abs(Ew - Ec))


@pytest.mark.parametrize("G", [1, 32])
def test_zundel_shape_T100_against_oracle(G):
    """The Zundel shape of BASELINE configs[4] with the 100 training states of the reference's learning curve: N = 28,
    AO slices 9,2,2,2,9,2,2, T = 100 -> 5050 pair rows (12.4 GB in the reference's pack2 layout, 3.3 GB compressed);
    one geometry per call and batches of 32, both layouts, against the oracle on the pack2 rows."""
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAOBatch, BatchedEvaluator
    from evcont_amd.synthetic import make_device_ao, make_device_trdm_rows
    n, A, T, sizes = 28, 7, 100, (9, 2, 2, 2, 9, 2, 2)
    dev = torch.device(DEV)
    S, one, rows = make_device_trdm_rows(n, T, 2, 4100, dev)
    aos = [make_device_ao(n, A, 4100000 + k, dev, sizes, ip1_rs_symmetric=True) for k in range(G)]
    slots = sorted({0, G // 2, G - 1})
    one_h, two_h, S_h = one.cpu().numpy(), rows.cpu().numpy(), S.cpu().numpy()
    want = {k: orc.energy_with_grad(_bundle_from_device(aos[k]), one_h, two_h, S_h) for k in slots}
    del two_h
    trd = DeviceTRDMs.from_device_rows(one, rows, S, 2)
    be = BatchedEvaluator(trd, A, G)
    got = {"pack2": be.energies_with_grads(DeviceAOBatch.stack(aos))}
    del be
    trd.compress_sym8_()
    del rows
    be = BatchedEvaluator(trd, A, G)
    got["sym8"] = be.energies_with_grads(DeviceAOBatch.stack([a.packed_ip1(eri=True) for a in aos]))
    for leg, (E, grad) in got.items():
        de = max(abs(E[k] - want[k][0]) for k in slots)
        dg = max(float(np.abs(grad[k] - want[k][1]).max()) for k in slots)
        assert de < 1e-10 and dg < 1e-9, (leg, de, dg)


@pytest.mark.parametrize("G", [1, 3])
def test_h2o_vtz_shape_n58_against_oracle(G):
    """cc-pVTZ water, the reference's largest orbital space (scripts/MD/H2O/md_H2O_vtz_CAS_continuation.py:31: N = 58,
    AO slices 30/14/14), T = 8: the full pipeline beyond the 32-orbital fast path, pack2 and sym8, against the oracle on
    the pack2 rows."""
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAOBatch, BatchedEvaluator
    from evcont_amd.synthetic import make_device_ao, make_device_trdm_rows
    n, A, T, sizes = 58, 3, 8, (30, 14, 14)
    dev = torch.device(DEV)
    S, one, rows = make_device_trdm_rows(n, T, 2, 4500, dev)
    aos = [make_device_ao(n, A, 4500000 + k, dev, sizes, ip1_rs_symmetric=True) for k in range(G)]
    slots = sorted({0, G - 1})
    one_h, two_h, S_h = one.cpu().numpy(), rows.cpu().numpy(), S.cpu().numpy()
    want = {k: orc.energy_with_grad(_bundle_from_device(aos[k]), one_h, two_h, S_h) for k in slots}
    del two_h
    trd = DeviceTRDMs.from_device_rows(one, rows, S, 2)
    got = {"pack2": BatchedEvaluator(trd, A, G).energies_with_grads(DeviceAOBatch.stack(aos))}
    trd.compress_sym8_()
    del rows
    got["sym8"] = BatchedEvaluator(trd, A, G).energies_with_grads(DeviceAOBatch.stack(aos))
    for leg, (E, grad) in got.items():
        de = max(abs(E[k] - want[k][0]) for k in slots)
        dg = max(float(np.abs(grad[k] - want[k][1]).max()) for k in slots)
        assert de < 1e-10 and dg < 1e-9, (leg, de, dg)


def test_leave_one_out_energies_T40():
    """The active-learning re-evaluations (MD_utils.py:264-299,448-483) at T = 40: the energies of every leave-one-out
    subset (39 states: per-problem overlap matrices through the large-T subspace kernel, in chunks) and of the full set,
    from ONE contraction per geometry, against scipy.linalg.eigh on the sub-matrices."""
    import scipy.linalg as sla
    from evcont_amd.active_learning import subset_energies
    T, B = 40, 3
    rng = np.random.default_rng(3801)
    A = rng.standard_normal((T, T))
    S = A @ A.T / T + np.eye(T)
    H = rng.standard_normal((B, T, T))
    H = 0.5 * (H + H.transpose(0, 2, 1)) - 3.0 * np.eye(T)
    enuc = rng.standard_normal(B)
    subsets = [[i for i in range(T) if i != j] for j in range(T)] + [list(range(T))]
    got = subset_energies(torch.from_numpy(H).to(DEV), torch.from_numpy(S).to(DEV), torch.from_numpy(enuc).to(DEV),
                          subsets).cpu().numpy()
    for b in range(B):
        for k, ids in enumerate(subsets):
            ix = np.ix_(ids, ids)
            want = sla.eigh(H[b][ix], S[ix], eigvals_only=True)[0] + enuc[b]
            assert abs(got[b, k] - want) < 1e-10, (b, k, got[b, k], want)


@pytest.mark.parametrize("n,sizes,T", [(70, (40, 30), 3), (96, (50, 46), 2)])
def test_more_than_64_orbitals_against_oracle(n, sizes, T):
    """Orbital spaces beyond 64 (the reference has no bound; its largest is cc-pVTZ water, N = 58): Loewdin on the
    1024-thread Jacobi with two work matrices in the workspace, quarter transforms up to 96 padded columns, Y2 in 64 x 64
    quadrants, gradient tail with its operands read through the caches.  pack2, sym8 and the unpacked full6 layout against
    the oracle."""
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAOBatch, BatchedEvaluator
    from evcont_amd.synthetic import make_device_ao, make_device_trdm_rows
    A, G = len(sizes), (2 if n <= 70 else 1)
    dev = torch.device(DEV)
    S, one, rows = make_device_trdm_rows(n, T, 2, 4700 + n, dev)
    aos = [make_device_ao(n, A, 4700000 + 10 * n + k, dev, sizes, ip1_rs_symmetric=True) for k in range(G)]
    one_h, two_h, S_h = one.cpu().numpy(), rows.cpu().numpy(), S.cpu().numpy()
    want = [orc.energy_with_grad(_bundle_from_device(a), one_h, two_h, S_h) for a in aos]
    trd = DeviceTRDMs.from_device_rows(one, rows, S, 2)
    got = {"pack2": BatchedEvaluator(trd, A, G).energies_with_grads(DeviceAOBatch.stack(aos))}
    if n <= 70:   # the (T,T,N^4) layout: through the unpacked path (sym_oao_t, y2_kernel)
        nn = n * n
        r, c = np.tril_indices(nn)
        full = np.zeros((T, T, nn, nn))
        a_, b_ = np.tril_indices(T)
        for p_, (ia, ib) in enumerate(zip(a_, b_)):
            m = np.zeros((nn, nn))
            m[r, c] = two_h[p_]
            m = m + m.T - np.diag(np.diag(m))
            full[ia, ib] = m
            full[ib, ia] = m.reshape(n, n, n, n).transpose(1, 0, 3, 2).reshape(nn, nn)
        # (bra <-> ket exchange of a pair goes with p <-> q, r <-> s; the oracle on THIS array is the reference here)
        full = full.reshape(T, T, n, n, n, n)
        want6 = [orc.energy_with_grad(_bundle_from_device(a), one_h, full, S_h) for a in aos]
        E6, g6 = BatchedEvaluator(DeviceTRDMs(one_h, full, S_h, dev), A, G).energies_with_grads(DeviceAOBatch.stack(aos))
        for k in range(G):
            assert abs(E6[k] - want6[k][0]) < 1e-9 and np.abs(g6[k] - want6[k][1]).max() < 1e-8, ("full6", k)
        del full
    del two_h
    trd.compress_sym8_()
    del rows
    got["sym8"] = BatchedEvaluator(trd, A, G).energies_with_grads(DeviceAOBatch.stack(aos))
    for leg, (E, grad) in got.items():
        de = max(abs(E[k] - want[k][0]) for k in range(G))
        dg = max(float(np.abs(grad[k] - want[k][1]).max()) for k in range(G))
        assert de < 1e-9 and dg < 1e-8, (leg, de, dg)


def test_reference_golden_T40():
    """T = 40 against vectors the REFERENCE itself produced (tests/golden/make_golden_large_T.py): energies, forces,
    predicted RDMs on the pack2 layout; energies and forces on the compressed one; the six lowest roots."""
    import os
    from conftest import GOLDEN_DIR
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAO, ContinuationEvaluator
    from evcont_amd.synthetic import AOArrays
    with np.load(os.path.join(GOLDEN_DIR, "largeT_n3t40a2.npz")) as z:
        g = {k: z[k] for k in z.files}
    ao = AOArrays(S=g["S"], hcore=g["hcore"], eri=g["eri"], ipovlp=g["ipovlp"], dhcore=g["dhcore"], eri_ip1=g["eri_ip1"],
                  aoslices=g["aoslices"], enuc=float(g["enuc"]), gnuc=g["gnuc"])
    A = int(g["aoslices"].shape[0])
    ev = ContinuationEvaluator(DeviceTRDMs(g["one_RDM"], g["two_RDM_pack2"], g["S_train"], DEV), A)
    E, grad, D, G = ev.energy_with_grad(DeviceAO.from_arrays(ao, DEV), True)
    assert abs(E - float(g["ewg_E_pack2"])) < 1e-10
    np.testing.assert_allclose(grad, g["ewg_grad_pack2"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(D, g["ewg_D_pack2"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(G, g["ewg_G_pack2"].reshape(G.shape), rtol=0, atol=1e-10)
    e6, _ = ev.energies(DeviceAO.from_arrays(ao, DEV, energy_only=True), nroots=6)       # > 4 roots: the Jacobi route
    np.testing.assert_allclose(e6, g["ms_E_pack2"] + float(g["enuc"]), rtol=0, atol=1e-10)
    e3, _ = ev.energies(DeviceAO.from_arrays(ao, DEV, energy_only=True), nroots=3)       # few-roots route
    np.testing.assert_allclose(e3, g["ms_E_pack2"][:3] + float(g["enuc"]), rtol=0, atol=1e-10)
    ev8 = ContinuationEvaluator(DeviceTRDMs(g["one_RDM"], g["two_RDM_pack2"], g["S_train"], DEV, compress="sym8"), A)
    E8, g8 = ev8.energy_with_grad(DeviceAO.from_arrays(ao, DEV, pack_ip1=True, pack_eri=True))
    assert abs(E8 - float(g["ewg_E_pack2"])) < 1e-10
    np.testing.assert_allclose(g8, g["ewg_grad_pack2"], rtol=0, atol=1e-9)
