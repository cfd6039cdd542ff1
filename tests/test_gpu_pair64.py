"""The symmetric pipeline for 32 < N <= 64 orbitals (csrc/pair64.hip: pt64_kernel, y2_64_kernel): compressed layout with
int2e / int2e_ip1 handed over packed (aosym s4 / s2kl, as PySCF delivers them), full pipeline against the oracle on the
ORIGINAL pack2 rows and full integral arrays (get_energy_with_grad, ab_initio_gradients_loewdin.py:308-379) -- at the
kernel boundaries (33: first size beyond the 32-orbital kernels; 48: three full tiles; 58: the reference's cc-pVTZ
water, md_H2O_vtz_CAS_continuation.py:25-33; 63, 64: odd / full last tile), single geometries (Newton-Schulz Loewdin
half) and batches, energy only, predicted RDMs, and against the quarter-step route the same library takes for full
integral arrays."""
import numpy as np
import pytest
import torch

from oracle import evcont_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _bundle(ao):
    c = lambda t: t.cpu().numpy()
    return orc.AOBundle(S=c(ao.S), hcore=c(ao.hcore), eri=c(ao.eri), ipovlp=c(ao.ipovlp), dhcore=c(ao.dhcore),
                        eri_ip1=c(ao.eri_ip1), aoslices=c(ao.aoslices), enuc=ao.enuc, gnuc=c(ao.gnuc))


def _setup(n, T, A, sizes, seed, G):
    from evcont_amd.evaluator import DeviceTRDMs
    from evcont_amd.synthetic import make_device_ao, make_device_trdm_rows
    dev = torch.device(DEV)
    S, one, rows = make_device_trdm_rows(n, T, 2, seed, dev)
    aos = [make_device_ao(n, A, 1000 * seed + k, dev, sizes, ip1_rs_symmetric=True) for k in range(G)]
    one_h, two_h, S_h = one.cpu().numpy(), rows.cpu().numpy(), S.cpu().numpy()
    trd = DeviceTRDMs.from_device_rows(one, rows, S, 2)
    trd.compress_sym8_()
    return trd, aos, (one_h, two_h, S_h)


# (G = 13: a batch beyond the split threshold -- one-kernel Loewdin step, K5 / K8 on the matrix cores)
@pytest.mark.parametrize("n,T,A,sizes,G", [(33, 3, 3, (11, 11, 11), 1), (40, 4, 2, (25, 15), 13), (48, 3, 3, (16, 16, 16), 4),
                                           (58, 8, 3, (30, 14, 14), 1), (58, 8, 3, (30, 14, 14), 4),
                                           (63, 2, 3, (21, 21, 21), 1), (64, 3, 2, (40, 24), 2)])
def test_pair64_pipeline_against_oracle(n, T, A, sizes, G):
    from evcont_amd import _lib
    from evcont_amd.evaluator import DeviceAOBatch, BatchedEvaluator
    trd, aos, (one_h, two_h, S_h) = _setup(n, T, A, sizes, 5100 + n, G)
    slots = sorted({0, G - 1})
    want = {k: orc.energy_with_grad(_bundle(aos[k]), one_h, two_h, S_h) for k in slots}
    be = BatchedEvaluator(trd, A, G)
    E, grad = be.energies_with_grads(DeviceAOBatch.stack([a.packed_ip1(eri=True) for a in aos]))
    assert _lib.load().evc_profile_kernel(2).decode().startswith("pt64_kernel")
    assert _lib.load().evc_profile_kernel(4).decode().startswith("y2_64_kernel")
    for k in slots:
        assert abs(E[k] - want[k][0]) < 1e-10, (n, k, E[k], want[k][0])
        assert np.abs(grad[k] - want[k][1]).max() < 1e-9, (n, k)
    # the same geometries with the full arrays: the quarter-step route of the same library
    E2, grad2 = BatchedEvaluator(trd, A, G).energies_with_grads(DeviceAOBatch.stack(aos))
    np.testing.assert_allclose(E2, E, rtol=0, atol=1e-10)
    np.testing.assert_allclose(grad2, grad, rtol=0, atol=1e-9)


@pytest.mark.parametrize("n", [37, 58])
def test_pair64_energy_only_and_predicted_rdms(n):
    """Energy-only calls (first two pair steps alone) and calls that return the predicted RDMs (the unpacked 2-RDM is
    written beside the dense form the pipeline reads) against the oracle."""
    from evcont_amd.evaluator import DeviceAO, ContinuationEvaluator
    T, A = 3, 2
    trd, aos, (one_h, two_h, S_h) = _setup(n, T, A, None, 5200 + n, 1)
    Eo, go = orc.energy_with_grad(_bundle(aos[0]), one_h, two_h, S_h)
    ev = ContinuationEvaluator(trd, A, want_two_rdm=True)
    ao = aos[0].packed_ip1(eri=True)
    e, _ = ev.energies(ao, nroots=1)
    assert abs(e[0] - Eo) < 1e-10
    E, g, D, Gm = ev.energy_with_grad(ao, True)
    assert abs(E - Eo) < 1e-10 and np.abs(g - go).max() < 1e-9
    # the predicted RDMs: the same from the evaluator that takes the full arrays (quarter-step route)
    E2, g2, D2, G2 = ContinuationEvaluator(trd, A, want_two_rdm=True).energy_with_grad(aos[0], True)
    np.testing.assert_allclose(D, D2, rtol=0, atol=1e-11)
    np.testing.assert_allclose(Gm, G2, rtol=0, atol=1e-11)
    np.testing.assert_allclose(g, g2, rtol=0, atol=1e-9)


def test_pair64_phase_calls_on_row_slices():
    """The three phase entry points on two row slices of the compressed set (emulated pair sharding, the collectives
    laid out as distributed.PairShardedContinuation does) at N = 34 with packed inputs: the gradient phase of every
    slice finds the first pair step's intermediate its energy phase left in the workspace; sum of the partial gradients
    = the fused call."""
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAOBatch, BatchedEvaluator
    from evcont_amd.distributed import shard_rows
    from evcont_amd.synthetic import make_device_ao, make_device_trdm_rows
    dev = torch.device(DEV)
    n, T, A, G, world = 34, 4, 2, 2, 2
    S, one, rows = make_device_trdm_rows(n, T, 2, 5300, dev)
    aos = [make_device_ao(n, A, 5300000 + k, dev, None, ip1_rs_symmetric=True).packed_ip1(eri=True) for k in range(G)]
    aob = DeviceAOBatch.stack(aos)
    trd = DeviceTRDMs.from_device_rows(one, rows.clone(), S, 2)
    trd.compress_sym8_()
    Eref, gref = BatchedEvaluator(trd, A, G).energies_with_grads(aob)
    nrows = T * (T + 1) // 2
    chunk = -(-nrows // world)
    evs, send = [], []
    for r in range(world):
        r0, r1 = shard_rows(nrows, world, r)
        t_r = DeviceTRDMs.from_device_rows(one, rows[r0:r1].contiguous(), S, 2, row_offset=r0, rows_total=nrows)
        t_r.compress_sym8_()
        evs.append(BatchedEvaluator(t_r, A, G))
        buf = torch.zeros((G, chunk), dtype=torch.float64, device=dev)
        evs[-1].phase_hamiltonian(aob, buf)
        send.append(buf)
    rows_all = torch.stack(send).permute(1, 0, 2).reshape(G, world * chunk).contiguous()
    total = torch.zeros_like(evs[0].grad)
    for r, ev in enumerate(evs):
        ev.phase_solve(aob, rows_all, 1)
        ev.phase_gradient(aob, partial_rank=(r != 0))
        total += ev.grad
    torch.cuda.synchronize()
    for ev in evs:
        np.testing.assert_allclose(ev.energy[:, 0].cpu().numpy(), Eref, rtol=0, atol=1e-11)
    np.testing.assert_allclose(total.cpu().numpy()[:, :A], gref, rtol=0, atol=1e-10)
