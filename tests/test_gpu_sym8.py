"""GPU tests of the 8-fold compressed t-RDM layout (EVC_LAYOUT_SYM8, include/evcont_hip.h): built on the
device from any of the reference's four layouts, it must give the reference's energies, coefficients and
forces (the oracle is run on the ORIGINAL, un-symmetrised t-RDMs) whenever the AO integrals carry the index
symmetries of real two-electron integrals -- on seeded tensors with those symmetries and on physical
hydrogen-chain integrals.  Tolerances as in test_gpu_parity.py."""
import numpy as np
import pytest
import torch

from evcont_amd.synthetic import make_ao_arrays, make_trdms, pack_rows
from oracle import evcont_oracle as orc

pytestmark = pytest.mark.gpu

LAYOUTS = {"full6": (False, False), "pair5": (True, False), "elec3": (False, True), "pack2": (True, True)}


def layout(two, name):
    p, e = LAYOUTS[name]
    return pack_rows(two, p, e) if (p or e) else two


def sym8(G):
    """Mean over the index permutations of real two-electron integrals (last four axes)."""
    a = G + np.swapaxes(G, -4, -3)
    a = a + np.swapaxes(a, -2, -1)
    a = a + np.moveaxis(a, (-2, -1), (-4, -3))
    return a / 8.0


def bundle(a):
    return orc.AOBundle(a.S, a.hcore, a.eri, a.ipovlp, a.dhcore, a.eri_ip1, a.aoslices, a.enuc, a.gnuc)


def test_layout_shape_and_column_order():
    """Column u(u+1)/2+v, u = i(i+1)/2+j: the compressed rows are the symmetrised tensor at those indices."""
    from evcont_amd.evaluator import DeviceTRDMs, layout_shape
    dev = torch.device("cuda:0")
    n, T = 5, 3
    S, one, two = make_trdms(n, T, 11)
    assert layout_shape(8, T, n) == (T * (T + 1) // 2, 15 * 16 // 2)
    ref = None
    for lname in LAYOUTS:
        t = DeviceTRDMs(one, layout(two, lname), S, dev, compress="sym8")
        assert t.layout == 8 and (t.rows_total, t.cols) == layout_shape(8, T, n) and t.ld % 16 == 0
        rows = t.two[:, : t.cols].cpu().numpy()
        if ref is None:
            gs = sym8(two)
            a, b = np.tril_indices(T)
            iu, ju = np.tril_indices(n)
            U, V = np.tril_indices(len(iu))
            ref = gs[a, b][:, iu[U], ju[U], iu[V], ju[V]]
        np.testing.assert_allclose(rows, ref, rtol=0, atol=1e-15)
        assert float(t.two[:, t.cols:].abs().max()) == 0.0


@pytest.mark.parametrize("n,T,A,lname", [
    (4, 2, 2, "full6"), (6, 3, 3, "pair5"), (7, 4, 2, "elec3"), (10, 5, 10, "pack2"),
    (17, 4, 3, "pack2"),       # pair transform <32>, ragged q tile
    (16, 3, 2, "pack2"),       # exactly one MFMA tile
    (30, 3, 5, "pack2"),       # N of the headline workload
    (31, 2, 3, "pack2"),       # odd N just below the pair-transform limit (dummy dimension in the eigensolver)
    (32, 2, 4, "pack2"),       # the pair-transform limit, no padding
    (34, 2, 2, "pack2"),       # N > 32: quarter steps + pack_sym8 kernel
])
def test_sym8_matches_reference_layouts(n, T, A, lname):
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAO, ContinuationEvaluator
    dev = torch.device("cuda:0")
    S, one, two = make_trdms(n, T, 40 + n)
    two_l = layout(two, lname)
    ao = make_ao_arrays(n, A, 90 + n, ip1_rs_symmetric=True)
    ev = ContinuationEvaluator(DeviceTRDMs(one, two_l, S, dev, compress="sym8"), A)
    E, g, D, G = ev.energy_with_grad(DeviceAO.from_arrays(ao, dev), return_density_matrices=True)
    Eo, go, Do, Go = orc.energy_with_grad(bundle(ao), one, two_l, S, True, True)
    assert abs(E - Eo) < 1e-10
    np.testing.assert_allclose(g, go, rtol=0, atol=1e-9)
    np.testing.assert_allclose(D, Do, rtol=0, atol=1e-11)
    np.testing.assert_allclose(G, sym8(np.asarray(Go).reshape(n, n, n, n)), rtol=0, atol=1e-11)
    # excited roots and coefficient vectors of the energy-only call
    nroots = min(T, 3)
    e, c = ev.energies(DeviceAO.from_arrays(ao, dev), nroots)
    X = orc.loewdin_trafo(ao.S)
    h1, h2 = orc.integrals_oao(bundle(ao), X)
    eo, co = orc.approximate_multistate(h1, h2, one, two_l, S, nroots)
    np.testing.assert_allclose(e, np.asarray(eo) + ao.enuc, rtol=0, atol=1e-10)


def test_sym8_needs_the_integral_symmetry(monkeypatch):
    """With a general (not r<->s symmetric) eri_ip1 the compressed layout is NOT exact.  By default the first call of
    an evaluator verifies the symmetries on the device and raises (EVCONT_AMD_CHECK_SYM=1); with the check disabled
    the documented behaviour shows: energies still agree, forces differ (the pipeline reads eri_ip1 and the
    intermediates of the two rotations in their lower triangles only)."""
    from evcont_amd import evaluator as evm
    from evcont_amd._lib import EvcontHipError
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAO, DeviceAOBatch, ContinuationEvaluator, BatchedEvaluator
    dev = torch.device("cuda:0")
    n, T, A = 6, 3, 3
    S, one, two = make_trdms(n, T, 5)
    ao = make_ao_arrays(n, A, 6)                       # general eri_ip1
    trd = DeviceTRDMs(one, two, S, dev, compress="sym8")
    with pytest.raises(EvcontHipError, match="eri_ip1"):
        ContinuationEvaluator(trd, A).energy_with_grad(DeviceAO.from_arrays(ao, dev))
    with pytest.raises(EvcontHipError, match="eri_ip1"):
        BatchedEvaluator(trd, A, 2).energies_with_grads(DeviceAOBatch.from_arrays([ao, ao], dev))
    monkeypatch.setattr(evm, "_host_checks_done", set())
    with pytest.raises(EvcontHipError, match="eri_ip1"):          # the host-side packing helper checks as well
        DeviceAO.from_arrays(ao, dev, pack_ip1=True, pack_eri=True)
    bad = make_ao_arrays(n, A, 7, ip1_rs_symmetric=True)
    bad.eri = bad.eri + 1e-3 * np.random.default_rng(1).standard_normal(bad.eri.shape)      # not 8-fold symmetric
    with pytest.raises(EvcontHipError, match="eri:"):
        ContinuationEvaluator(trd, A).energy_with_grad(DeviceAO.from_arrays(bad, dev))
    # energy-only calls do not look at eri_ip1
    ContinuationEvaluator(trd, A).energies(DeviceAO.from_arrays(ao, dev, energy_only=True), 1)
    monkeypatch.setattr(evm, "_CHECK_SYM", "0")
    ev = ContinuationEvaluator(trd, A)
    E, g = ev.energy_with_grad(DeviceAO.from_arrays(ao, dev))
    Eo, go = orc.energy_with_grad(bundle(ao), one, two, S)
    assert abs(E - Eo) < 1e-10
    assert np.abs(g - go).max() > 1e-4
    monkeypatch.setattr(evm, "_CHECK_SYM", "2")
    # symmetrising eri_ip1 in its last two indices restores the agreement (and passes the check on every call)
    ao.eri_ip1 = np.ascontiguousarray(0.5 * (ao.eri_ip1 + ao.eri_ip1.transpose(0, 1, 2, 4, 3)))
    E, g = ev.energy_with_grad(DeviceAO.from_arrays(ao, dev))
    Eo, go = orc.energy_with_grad(bundle(ao), one, two, S)
    assert abs(E - Eo) < 1e-10
    np.testing.assert_allclose(g, go, rtol=0, atol=1e-9)


@pytest.mark.parametrize("n,T,A,G", [(10, 5, 10, 3), (18, 6, 3, 9), (21, 5, 3, 17), (12, 4, 4, 33)])
def test_sym8_batched(n, T, A, G):
    """Batched evaluator on the compressed layout (VALU, wave-rows and matrix-core streaming kernels)."""
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAO, DeviceAOBatch, BatchedEvaluator
    dev = torch.device("cuda:0")
    S, one, two = make_trdms(n, T, 300 + n)
    two_l = pack_rows(two, True, True)
    aos = [make_ao_arrays(n, A, 800 + k, ip1_rs_symmetric=True) for k in range(G)]
    be = BatchedEvaluator(DeviceTRDMs(one, two_l, S, dev, compress="sym8"), A, G, keep_density_matrices=True)
    E, grad = be.energies_with_grads(DeviceAOBatch.stack([DeviceAO.from_arrays(a, dev) for a in aos]))
    for k in sorted({0, G // 2, G - 1}):
        Eo, go, Do, Go = orc.energy_with_grad(bundle(aos[k]), one, two_l, S, True, True)
        assert abs(E[k] - Eo) < 1e-9, k
        np.testing.assert_allclose(grad[k], go, rtol=0, atol=1e-8)
        np.testing.assert_allclose(be.g_pred[k].cpu().numpy(), sym8(np.asarray(Go).reshape(n, n, n, n)), rtol=0,
                                   atol=1e-10)


def test_sym8_pair_sharded_rows():
    """Row-sharded compressed sets (pair sharding): shards of the pack2 rows compress to the
    corresponding shards of the compressed rows."""
    from evcont_amd.evaluator import DeviceTRDMs
    dev = torch.device("cuda:0")
    n, T = 8, 5
    S, one, two = make_trdms(n, T, 77)
    two_l = pack_rows(two, True, True)
    full = DeviceTRDMs(one, two_l, S, dev, compress="sym8")
    P = T * (T + 1) // 2
    for r0, r1 in ((0, 7), (7, P), (4, 4)):
        part = DeviceTRDMs(one, two_l, S, dev, row_range=(r0, r1), compress="sym8")
        assert (part.row_offset, part.rows_local, part.rows_total) == (r0, r1 - r0, P)
        if r1 > r0:
            assert torch.equal(part.two[: r1 - r0], full.two[r0:r1])


def test_sym8_h10_physical(h10_fci):
    """configs[1] on physical integrals: H10 / STO-3G / 5 FCI training states, compressed layout vs the CPU
    path on the stored (pair-packed) t-RDMs, at training points (exact FCI energies) and at a bent geometry."""
    from evcont_amd.hchain import s_gaussian_mol, hydrogen_chain
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAO, ContinuationEvaluator
    dev = torch.device("cuda:0")
    h10 = h10_fci
    S, one, two = h10["overlap"], h10["one_rdm"], h10["two_rdm_pack2"]
    ev = ContinuationEvaluator(DeviceTRDMs(one, two, S, dev, compress="sym8"), 10)
    mols = [hydrogen_chain(10, float(d)) for d in h10["spacings"]] + [s_gaussian_mol(h10["R_test"])]
    for k, m in enumerate(mols):
        E, g = ev.energy_with_grad(DeviceAO.from_arrays(m, dev))
        Eo, go = orc.energy_with_grad(bundle(m), one, two, S)
        assert abs(E - Eo) < 1e-10 and np.abs(g - go).max() < 1e-9
        if k < len(h10["ens"]):
            assert abs(E - float(h10["ens"][k])) < 1e-8


def test_sym8_h30_full_size_against_pack2():
    """Headline workload at FULL size (N=30, A=30, T=20): the compressed set built from the resident pack2 rows
    reproduces the pack2 pipeline (itself pinned to the reference at small sizes) on integrals with the
    symmetries of real ones; batch of 16 through the matrix-core kernels and single geometries."""
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAOBatch, ContinuationEvaluator, BatchedEvaluator
    from evcont_amd.synthetic import make_device_ao, make_device_trdm_rows
    dev = torch.device("cuda:0")
    n, A, T, G = 30, 30, 20, 16
    S, one, rows = make_device_trdm_rows(n, T, 2, 1236, dev)
    trd = DeviceTRDMs.from_device_rows(one, rows, S, 2)
    aos = [make_device_ao(n, A, 5000 + k, dev, ip1_rs_symmetric=True) for k in range(G)]
    ref = ContinuationEvaluator(trd, A)
    want = {k: ref.energy_with_grad(aos[k]) for k in (0, 5, 15)}
    del ref
    trd8 = DeviceTRDMs.from_device_rows(one, rows, S, 2).compress_sym8_()
    del rows, trd
    assert trd8.cols == 108345 and trd8.two.shape == (210, 108352)
    be = BatchedEvaluator(trd8, A, G)
    be.enqueue(DeviceAOBatch.stack(aos))
    be.synchronize()
    E, grad = be.energy[:, 0].cpu().numpy(), be.grad.cpu().numpy()
    single = ContinuationEvaluator(trd8, A)
    for k, (Ep, gp) in want.items():
        assert abs(E[k] - Ep) < 1e-9, (k, E[k], Ep)
        np.testing.assert_allclose(grad[k], gp, rtol=0, atol=1e-8)
        Es, gs = single.energy_with_grad(aos[k])
        assert abs(Es - Ep) < 1e-9
        np.testing.assert_allclose(gs, gp, rtol=0, atol=1e-8)


def test_sym8_through_the_reference_api(h10_fci, tmp_path):
    """``set_trdm_compression("sym8")`` switches the mol-level entry points (get_energy_with_grad, the
    *_OAO energies, the MD scanner / trajectory driver, the container's device copy) to the compressed set
    unconditionally (the default "auto" does so only where the caller cannot tell the difference); results on
    physical integrals are those of the caller's layout (``None``)."""
    from evcont_amd import ab_initio_eigenvector_continuation as aec
    from evcont_amd.ab_initio_gradients_loewdin import get_energy_with_grad
    from evcont_amd.MD_utils import get_trajectory
    from evcont_amd.hchain import s_gaussian_mol, hydrogen_chain
    from evcont_amd import cache
    h10 = h10_fci
    S, one, two = h10["overlap"], h10["one_rdm"], h10["two_rdm_pack2"]
    m = s_gaussian_mol(h10["R_test"])
    assert aec.get_trdm_compression() == "auto"
    try:
        aec.set_trdm_compression(None)
        E0, g0 = get_energy_with_grad(m, one, two, S)
        e0, c0 = aec.approximate_multistate_OAO(m, one, two, S, nroots=3)
        traj0 = get_trajectory(hydrogen_chain(10, 1.9), S, one, two, dt=5.0, steps=4)
        aec.set_trdm_compression("auto")
        Ea, ga = get_energy_with_grad(m, one, two, S)
        traja = get_trajectory(hydrogen_chain(10, 1.9), S, one, two, dt=5.0, steps=4)
        aec.set_trdm_compression("sym8")
        E1, g1, D1, G1 = get_energy_with_grad(m, one, two, S, return_density_matrices=True)
        e1, c1 = aec.approximate_multistate_OAO(m, one, two, S, nroots=3)
        traj1 = get_trajectory(hydrogen_chain(10, 1.9), S, one, two, dt=5.0, steps=4)
        # hermitian=False keeps working on the layout the caller passed
        en, _ = aec.approximate_ground_state_OAO(m, one, two, S, hermitian=False)
    finally:
        aec.set_trdm_compression("auto")
        cache.clear()
    assert abs(Ea - E0) < 1e-10 and np.abs(ga - g0).max() < 1e-9
    np.testing.assert_allclose(traja, traj0, rtol=0, atol=1e-8)
    assert abs(E1 - E0) < 1e-10 and np.abs(g1 - g0).max() < 1e-9
    np.testing.assert_allclose(e1, e0, rtol=0, atol=1e-10)
    np.testing.assert_allclose(traj1, traj0, rtol=0, atol=1e-8)
    assert abs(en - E0) < 1e-8
    np.testing.assert_allclose(G1, sym8(G1), rtol=0, atol=1e-13)       # the stored 2-RDM is the symmetrised one
    with pytest.raises(ValueError):
        aec.set_trdm_compression("sym4")


# (sizes around the shapes of the pair-transform kernels: fewer pairs than one 8-pair tile (n = 2, 3), the 16/17 and 30/31
# boundaries -- the software-pipelined kernel takes n <= 30, n = 31, 32 the phase-alternating one --, odd and even n(n+1)/2)
@pytest.mark.parametrize("n,T,A", [(2, 2, 1), (3, 2, 2), (5, 3, 2), (7, 3, 3), (10, 5, 10), (16, 3, 2), (17, 3, 2),
                                   (18, 4, 3), (29, 2, 3), (30, 3, 5), (31, 2, 3), (32, 2, 4)])
def test_packed_ip1_input(n, T, A):
    """EVC_FLAG_IP1_S2KL: int2e_ip1 handed over packed in its last two AO indices (PySCF aosym="s2kl"), host-packed
    and device-gathered, single and batched, against the oracle on the full arrays and the original t-RDMs."""
    from evcont_amd.evaluator import (DeviceTRDMs, DeviceAO, DeviceAOBatch, ContinuationEvaluator, BatchedEvaluator)
    dev = torch.device("cuda:0")
    S, one, two = make_trdms(n, T, 140 + n)
    two_l = pack_rows(two, True, True)
    aos = [make_ao_arrays(n, A, 190 + n + k, ip1_rs_symmetric=True) for k in range(3)]
    trd = DeviceTRDMs(one, two_l, S, dev, compress="sym8")
    ev = ContinuationEvaluator(trd, A)
    dao = DeviceAO.from_arrays(aos[0], dev, pack_ip1=True)
    assert dao.ip1_s2kl and tuple(dao.eri_ip1.shape) == (3, n, n, n * (n + 1) // 2)
    assert torch.equal(DeviceAO.from_arrays(aos[0], dev).packed_ip1().eri_ip1, dao.eri_ip1)
    E, g = ev.energy_with_grad(dao)
    Eo, go = orc.energy_with_grad(bundle(aos[0]), one, two_l, S)
    assert abs(E - Eo) < 1e-10
    np.testing.assert_allclose(g, go, rtol=0, atol=1e-9)
    be = BatchedEvaluator(trd, A, 3)
    Eb, gb = be.energies_with_grads(DeviceAOBatch.from_arrays(aos, dev, pack_ip1=True))
    for k in range(3):
        Eo, go = orc.energy_with_grad(bundle(aos[k]), one, two_l, S)
        assert abs(Eb[k] - Eo) < 1e-9
        np.testing.assert_allclose(gb[k], go, rtol=0, atol=1e-8)
    # int2e packed in both index pairs as well (aosym="s4"): host-packed and device-gathered, energy-only too
    dao4 = DeviceAO.from_arrays(aos[0], dev, pack_ip1=True, pack_eri=True)
    npr = n * (n + 1) // 2
    assert dao4.eri_s4 and tuple(dao4.eri.shape) == (npr, npr)
    assert torch.equal(DeviceAO.from_arrays(aos[0], dev).packed_ip1(eri=True).eri, dao4.eri)
    E4, g4 = ev.energy_with_grad(dao4)
    Eo, go = orc.energy_with_grad(bundle(aos[0]), one, two_l, S)
    assert abs(E4 - Eo) < 1e-10
    np.testing.assert_allclose(g4, go, rtol=0, atol=1e-9)
    e4, _ = ev.energies(DeviceAO.from_arrays(aos[0], dev, energy_only=True, pack_eri=True), 1)
    assert abs(e4[0] - Eo) < 1e-10
    Eb4, gb4 = be.energies_with_grads(DeviceAOBatch.from_arrays(aos, dev, pack_ip1=True, pack_eri=True))
    np.testing.assert_allclose(Eb4, Eb, rtol=0, atol=1e-11)
    np.testing.assert_allclose(gb4, gb, rtol=0, atol=1e-10)
    # the packed form is only understood by the symmetric pipeline of the compressed layout
    plain = ContinuationEvaluator(DeviceTRDMs(one, two_l, S, dev), A)
    from evcont_amd._lib import EvcontHipError
    with pytest.raises(EvcontHipError):
        plain.energy_with_grad(dao)
