"""GPU tests of EVC_FLAG_WARM_START: the Jacobi eigensolvers (overlap matrix, subspace problem) started
from the previous call's eigenvectors must reproduce the cold-start results to solver tolerance, along a
slowly varying sequence of geometries, across a jump to an unrelated geometry, and from a workspace whose
stored eigenvectors are stale, zero or NaN (detected, cold start)."""
import numpy as np
import pytest
import torch

from evcont_amd.synthetic import AOArrays, make_ao_arrays, make_trdms, pack_rows

pytestmark = pytest.mark.gpu


def _trdms(n, T, seed, dev, layout_packed=True):
    """Training set on the device; the large shapes are generated THERE (a (40, 40, 35^4) host array is 10 GB and
    40 s of numpy), the small ones on the host like everywhere else."""
    from evcont_amd.evaluator import DeviceTRDMs
    if n >= 32 and layout_packed:
        from evcont_amd.synthetic import make_device_trdm_rows
        S, one, rows = make_device_trdm_rows(n, T, 2, seed, dev)
        return DeviceTRDMs.from_device_rows(one, rows, S, 2)
    S, one, two = make_trdms(n, T, seed)
    return DeviceTRDMs(one, pack_rows(two, True, True) if layout_packed else two, S, dev)


def blend(a0: AOArrays, a1: AOArrays, t: float) -> AOArrays:
    mix = lambda x, y: (1.0 - t) * np.asarray(x) + t * np.asarray(y)
    return AOArrays(mix(a0.S, a1.S), mix(a0.hcore, a1.hcore), mix(a0.eri, a1.eri), mix(a0.ipovlp, a1.ipovlp),
                    mix(a0.dhcore, a1.dhcore), mix(a0.eri_ip1, a1.eri_ip1), a0.aoslices,
                    float(mix(a0.enuc, a1.enuc)), mix(a0.gnuc, a1.gnuc))


@pytest.mark.parametrize("n,T,A,lname", [(13, 5, 3, "pack2"), (30, 6, 30, "pack2"), (8, 4, 2, "full6"),
                                          (37, 40, 3, "pack2")])   # large-n Loewdin + large-T subspace kernels
def test_warm_start_matches_cold_start(n, T, A, lname):
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAO, ContinuationEvaluator
    dev = torch.device("cuda:0")
    trd = _trdms(n, T, 70 + n, dev, lname == "pack2")
    cold = ContinuationEvaluator(trd, A)
    warm = ContinuationEvaluator(trd, A, warm_start=True)
    a0, a1, other = make_ao_arrays(n, A, 1), make_ao_arrays(n, A, 2), make_ao_arrays(n, A, 3)
    # a slowly varying "trajectory", then a jump to an unrelated geometry and back
    seq = [blend(a0, a1, 0.002 * k) for k in range(5)] + [other, blend(a0, a1, 0.01)]
    for k, ao in enumerate(seq):
        dao = DeviceAO.from_arrays(ao, dev)
        Ec, gc, Dc, Gc = cold.energy_with_grad(dao, True)
        Ew, gw, Dw, Gw = warm.energy_with_grad(dao, True)
        assert abs(Ew - Ec) < 1e-11, (k, Ew, Ec)
        np.testing.assert_allclose(gw, gc, rtol=0, atol=1e-10)
        np.testing.assert_allclose(Dw, Dc, rtol=0, atol=1e-11)
        np.testing.assert_allclose(Gw, Gc, rtol=0, atol=1e-11)
        ew, cw = warm.energies(dao, nroots=min(3, T))
        ec, cc = cold.energies(dao, nroots=min(3, T))
        np.testing.assert_allclose(ew, ec, rtol=0, atol=1e-11)


@pytest.mark.parametrize("n,T", [(9, 4), (35, 40)])
@pytest.mark.parametrize("fill", ["zeros", "nan", "random"])
def test_warm_flag_on_stale_workspace_falls_back(fill, n, T):
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAO, ContinuationEvaluator
    dev = torch.device("cuda:0")
    A = 3
    trd = _trdms(n, T, 5, dev)
    dao = DeviceAO.from_arrays(make_ao_arrays(n, A, 11), dev)
    Ec, gc = ContinuationEvaluator(trd, A).energy_with_grad(dao)
    ev = ContinuationEvaluator(trd, A, warm_start=True)
    if fill == "zeros":
        ev.ws.zero_()
    elif fill == "nan":
        ev.ws.view(torch.float64).fill_(float("nan"))
    else:
        ev.ws.view(torch.float64).copy_(torch.randn(ev.ws.numel() // 8, dtype=torch.float64, device=dev))
    ev._primed = True                      # claim a previous call that never happened
    Ew, gw = ev.energy_with_grad(dao)
    assert abs(Ew - Ec) < 1e-11
    np.testing.assert_allclose(gw, gc, rtol=0, atol=1e-10)


@pytest.mark.parametrize("n,T,A", [(13, 5, 3), (10, 20, 4), (6, 70, 2)])
def test_cached_overlap_factorisation_follows_the_training_set(n, T, A):
    """The inverse Cholesky factor of S_train is cached in the workspace (it does not depend on the geometry) next to
    the matrix it was computed from: a workspace that is reused with ANOTHER training set of the same shape (the C ABI
    allows it) must notice and refactorise -- results identical to a fresh workspace's, in both directions, single
    and batched."""
    from evcont_amd.evaluator import (DeviceTRDMs, DeviceAO, DeviceAOBatch, ContinuationEvaluator, BatchedEvaluator)
    dev = torch.device("cuda:0")
    sets = []
    for seed in (5, 6):
        S, one, two = make_trdms(n, T, 300 + seed)
        sets.append(DeviceTRDMs(one, pack_rows(two, True, True), S, dev))
    aos = [make_ao_arrays(n, A, 40 + k) for k in range(3)]
    dao = DeviceAO.from_arrays(aos[0], dev)
    fresh = [ContinuationEvaluator(t, A).energy_with_grad(dao) for t in sets]
    ev = ContinuationEvaluator(sets[0], A)
    for k in (0, 0, 1, 1, 0):      # second call on a set: served from the cache; after a switch: a miss
        ev.t = sets[k]
        E, g = ev.energy_with_grad(dao)
        assert abs(E - fresh[k][0]) < 1e-12, (k, E, fresh[k][0])
        np.testing.assert_allclose(g, fresh[k][1], rtol=0, atol=1e-11)
    aob = DeviceAOBatch.from_arrays(aos, dev)
    freshb = [BatchedEvaluator(t, A, 3).energies_with_grads(aob) for t in sets]
    be = BatchedEvaluator(sets[1], A, 3)
    for k in (1, 1, 0, 0):
        be.t = sets[k]
        Eb, gb = be.energies_with_grads(aob)
        np.testing.assert_allclose(Eb, freshb[k][0], rtol=0, atol=1e-12)
        np.testing.assert_allclose(gb, freshb[k][1], rtol=0, atol=1e-11)
