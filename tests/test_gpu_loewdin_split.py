"""The Loewdin step in two halves (csrc/pipeline.hip "the Loewdin step in two halves", csrc/dense_small.hip loewdin_ns):
full calls compute X = S^-1/2 by a Newton-Schulz iteration on the matrix cores and run the eigendecomposition of S
(needed by the response term only, ab_initio_gradients_loewdin.py:41-134) off the critical path -- in the launch of
the subspace solve (N <= 32, T <= 32) or, for a few geometries of 33 ... 64 orbitals, on a side stream.
The phase calls keep the one-kernel form (eigensolver for everything): the two routes must agree, and both with the
oracle, for every matrix size, for well and badly conditioned overlap matrices (where the iteration must decline and
the kernel falls through to the eigensolver), single geometries and small batches, repeated calls on one workspace."""
import os

import numpy as np
import pytest
import torch

from evcont_amd.synthetic import make_ao_arrays, make_trdms, pack_rows
from oracle import evcont_oracle as orc

pytestmark = pytest.mark.gpu


def _with_overlap(ao, cond, seed):
    """The same bundle with S replaced by a symmetric positive definite matrix of the given condition number."""
    n = ao.S.shape[0]
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.logspace(0.0, -np.log10(cond), n) if n > 1 else np.ones(1)
    S = (q * lam) @ q.T
    ao.S = 0.5 * (S + S.T) * 1.7      # (norm away from 1: the scaling of the iteration is exercised)
    return ao


def _oracle(ao, one, two_p, S):
    b = orc.AOBundle(ao.S, ao.hcore, ao.eri, ao.ipovlp, ao.dhcore, ao.eri_ip1, ao.aoslices, ao.enuc, ao.gnuc)
    return orc.energy_with_grad(b, one, two_p, S)


@pytest.mark.parametrize("n,cond", [(n, c) for n in (1, 2, 3, 5, 8, 13, 16, 17, 24, 30, 31, 32) for c in (3.0, 1e3, 1e6)] +
                         # 32 < n <= 64: the 64 x 64 iteration (loewdin_ns64_kernel) in front of loewdin_big_kernel
                         # (63 / 64: tests/test_gpu_pair64.py, whose single-geometry cases take the same split route)
                         [(33, 3.0), (40, 1e3), (48, 1e6), (58, 1e3)])
def test_split_call_equals_phase_calls_and_oracle(n, cond):
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAO, ContinuationEvaluator
    dev = torch.device("cuda:0")
    T, A = 3, min(n, 3)
    S, one, two = make_trdms(n, T, 40 + n)
    two_p = pack_rows(two, True, True)
    ev = ContinuationEvaluator(DeviceTRDMs(one, two_p, S, dev), A)
    ref = ContinuationEvaluator(DeviceTRDMs(one, two_p, S, dev), A)
    scale = max(1.0, cond ** 0.5)     # |X| grows like cond^1/2: so do the rounding errors of everything built on it
    for rep in range(3 if n <= 32 else 2):   # (the same workspace again: fork / join events reused, U of the previous call consumed)
        ao = _with_overlap(make_ao_arrays(n, A, 900 + 7 * n + rep), cond, n + rep)
        dao = DeviceAO.from_arrays(ao, dev)
        E, g = ev.energy_with_grad(dao)                                  # split form (fewer than 12 geometries)
        rows = ref.phase_hamiltonian(dao)                                # one-kernel form
        ref.phase_solve(dao, rows.clone(), 1)
        ref.phase_gradient(dao, False)
        ref.synchronize()
        E2, g2 = float(ref.energy[0].item()), ref.grad[:A].cpu().numpy()
        Eo, go = _oracle(ao, one, two_p, S)
        gs = max(1.0, float(np.abs(g2).max()))
        assert abs(E - E2) < 1e-11 * scale * max(1.0, abs(Eo)), (n, cond, rep, E, E2)
        assert np.abs(g - g2).max() < 1e-10 * scale * gs, (n, cond, rep)
        assert abs(E - Eo) < 1e-10 * scale * max(1.0, abs(Eo)), (n, cond, rep, E, Eo)
        # (forces against the oracle only while the eigenvalues of S stay apart at five decimals: the reference buckets
        #  them with np.round(vals, 5), ab_initio_gradients_loewdin.py:55-56 -- DESIGN.md "known deviations" 1)
        if cond <= 1e3:
            assert np.abs(g - go).max() < 1e-9 * scale * gs, (n, cond, rep)


@pytest.mark.parametrize("n,cond", [(6, 1e10), (30, 1e10), (30, 1e12), (40, 1e10)])
def test_badly_conditioned_overlap_takes_the_eigensolver(n, cond):
    """cond(S) beyond what the iteration resolves: it declines (its residual test) and the same launch runs the
    eigensolver -- the result equals the one-kernel route to the accuracy such a matrix allows at all."""
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAO, ContinuationEvaluator
    dev = torch.device("cuda:0")
    T, A = 3, 2
    S, one, two = make_trdms(n, T, 50 + n)
    two_p = pack_rows(two, True, True)
    ev = ContinuationEvaluator(DeviceTRDMs(one, two_p, S, dev), A)
    ref = ContinuationEvaluator(DeviceTRDMs(one, two_p, S, dev), A)
    ao = _with_overlap(make_ao_arrays(n, A, 77 + n), cond, 5)
    dao = DeviceAO.from_arrays(ao, dev)
    E, g = ev.energy_with_grad(dao)
    rows = ref.phase_hamiltonian(dao)
    ref.phase_solve(dao, rows.clone(), 1)
    ref.phase_gradient(dao, False)
    ref.synchronize()
    E2, g2 = float(ref.energy[0].item()), ref.grad[:A].cpu().numpy()
    assert np.isfinite(E) and np.all(np.isfinite(g))
    # both routes ran the same eigensolver on the same matrix
    assert abs(E - E2) <= 1e-9 * max(1.0, abs(E2)), (E, E2)
    assert np.abs(g - g2).max() <= 1e-8 * max(1.0, float(np.abs(g2).max()))


@pytest.mark.parametrize("G", [2, 4, 11])
def test_small_batches_take_the_split_form(G):
    """Batches below the threshold (12 geometries) on the compressed layout with packed inputs against the oracle."""
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAOBatch, BatchedEvaluator
    dev = torch.device("cuda:0")
    n, A, T = 18, 4, 4
    S, one, two = make_trdms(n, T, 61)
    two_p = pack_rows(two, True, True)
    aos = [make_ao_arrays(n, A, 3000 + k, ip1_rs_symmetric=True) for k in range(G)]
    be = BatchedEvaluator(DeviceTRDMs(one, two_p, S, dev, compress="sym8"), A, G)
    for rep in range(2):
        E, grad = be.energies_with_grads(DeviceAOBatch.from_arrays(aos, dev, pack_ip1=True, pack_eri=True))
        for k in (0, G - 1):
            Eo, go = _oracle(aos[k], one, two_p, S)
            assert abs(E[k] - Eo) < 1e-10, (G, k)
            np.testing.assert_allclose(grad[k], go, rtol=0, atol=1e-9)


@pytest.mark.parametrize("n", [9, 30, 45])
def test_energy_only_call_then_gradient_phase(n):
    """EVC_FLAG_ENERGY_ONLY followed by evc_phase_gradient on the same workspace (what the hosted MD step does between
    its two uploads, evcont_amd/hosted.py): the gradient phase finds U and s of the split Loewdin step -- joined there,
    not in the call that forked the eigensolver -- and equals the fused call."""
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAO, ContinuationEvaluator
    dev = torch.device("cuda:0")
    T, A = 3, 3
    S, one, two = make_trdms(n, T, 70 + n)
    two_p = pack_rows(two, True, True)
    ev = ContinuationEvaluator(DeviceTRDMs(one, two_p, S, dev), A)
    ref = ContinuationEvaluator(DeviceTRDMs(one, two_p, S, dev), A)
    for rep in range(3):
        ao = make_ao_arrays(n, A, 500 + n + rep)
        dao = DeviceAO.from_arrays(ao, dev)
        e, _ = ev.energies(dao, nroots=1)            # energy only
        ev.phase_gradient(dao, False)
        ev.synchronize()
        g = ev.grad[:A].cpu().numpy()
        E2, g2 = ref.energy_with_grad(dao)
        Eo, go = _oracle(ao, one, two_p, S)
        assert abs(e[0] - E2) < 1e-12 and abs(E2 - Eo) < 1e-10
        np.testing.assert_allclose(g, g2, rtol=0, atol=1e-11)
        np.testing.assert_allclose(g, go, rtol=0, atol=1e-9)
