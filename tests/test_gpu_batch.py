"""GPU tests of the batched pipeline (count geometries per call, t-RDM streamed once per 8):
must reproduce the single-geometry path, which is itself pinned to the reference."""
import numpy as np
import pytest
import torch

from conftest import ao_from_golden
from evcont_amd.synthetic import make_ao_arrays, make_trdms, pack_rows

pytestmark = pytest.mark.gpu
LAYOUTS = {"full6": (False, False), "pair5": (True, False), "elec3": (False, True), "pack2": (True, True)}


def layout(two, name):
    p, e = LAYOUTS[name]
    return pack_rows(two, p, e) if (p or e) else two


@pytest.mark.parametrize("lname", list(LAYOUTS))
@pytest.mark.parametrize("G", [1, 2, 3, 8, 11])
def test_batch_matches_single(lname, G):
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAO, DeviceAOBatch, ContinuationEvaluator, BatchedEvaluator
    dev = torch.device("cuda:0")
    n, T, A = 7, 4, 3
    S, one, two = make_trdms(n, T, 31)
    t = DeviceTRDMs(one, layout(two, lname), S, dev)
    aos = [DeviceAO.from_arrays(make_ao_arrays(n, A, 500 + k, ao_sizes=(3, 2, 2)), dev) for k in range(G)]
    single = ContinuationEvaluator(t, A)
    ref = [single.energy_with_grad(a, return_density_matrices=True) for a in aos]
    be = BatchedEvaluator(t, A, G, keep_density_matrices=True)
    E, grad = be.energies_with_grads(DeviceAOBatch.stack(aos))
    for k in range(G):
        assert abs(E[k] - ref[k][0]) < 1e-11
        np.testing.assert_allclose(grad[k], ref[k][1], rtol=0, atol=1e-10)
        np.testing.assert_allclose(be.d_pred[k].cpu().numpy(), ref[k][2], rtol=0, atol=1e-11)
        np.testing.assert_allclose(be.g_pred[k].cpu().numpy(), ref[k][3], rtol=0, atol=1e-11)
    # energy-only / multistate through the batch entry
    be.enqueue(DeviceAOBatch.stack(aos), nroots=2, energy_only=True)
    be.synchronize()
    for k in range(G):
        es, _ = single.energies(aos[k], 2)
        np.testing.assert_allclose(be.energy[k, :2].cpu().numpy(), es, rtol=0, atol=1e-11)


def test_batch_golden(load_golden):
    """Batch of identical + different geometries against the reference's golden outputs."""
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAO, DeviceAOBatch, BatchedEvaluator
    dev = torch.device("cuda:0")
    g = load_golden("n6t3a3")
    ao = DeviceAO.from_arrays(ao_from_golden(g), dev)
    other = DeviceAO.from_arrays(make_ao_arrays(6, 3, 9), dev)
    t = DeviceTRDMs(g["one_RDM"], pack_rows(g["two_RDM"], True, True), g["S_train"], dev)
    be = BatchedEvaluator(t, 3, 5)
    E, grad = be.energies_with_grads(DeviceAOBatch.stack([ao, other, ao, other, ao]))
    for k in (0, 2, 4):
        assert abs(E[k] - float(g["ewg_E_pack2"])) < 1e-10
        np.testing.assert_allclose(grad[k], g["ewg_grad_pack2"], rtol=0, atol=1e-9)
    assert abs(E[1] - E[3]) < 1e-13 and abs(E[0] - E[1]) > 1e-6


@pytest.mark.parametrize("lname", ["pack2", "full6"])
@pytest.mark.parametrize("world,G", [(2, 3), (3, 16)])
def test_batched_phase_api_emulated_pair_sharding(lname, world, G):
    """Batched three-phase entry points on row slices (one BatchedEvaluator per emulated rank); the two
    collectives are emulated on one device exactly as distributed.PairShardedContinuation lays them out:
    (G, chunk) send buffers -> (world, G, chunk) -> (G, world*chunk); partial gradients summed."""
    from evcont_amd.evaluator import (DeviceTRDMs, DeviceAO, DeviceAOBatch, BatchedEvaluator, layout_shape)
    from evcont_amd.distributed import shard_rows
    dev = torch.device("cuda:0")
    n, T, A = 6, 4, 3
    S, one, two = make_trdms(n, T, 77)
    two_l = layout(two, lname)
    rows, _ = layout_shape(two_l.ndim, T, n)
    aob = DeviceAOBatch.stack([DeviceAO.from_arrays(make_ao_arrays(n, A, 900 + k), dev) for k in range(G)])
    full = BatchedEvaluator(DeviceTRDMs(one, two_l, S, dev), A, G)
    Eref, gref = full.energies_with_grads(aob)
    chunk = -(-rows // world)
    evs, send = [], []
    for r in range(world):
        r0, r1 = shard_rows(rows, world, r)
        evs.append(BatchedEvaluator(DeviceTRDMs(one, two_l, S, dev, row_range=(r0, r1)), A, G))
        buf = torch.zeros((G, chunk), dtype=torch.float64, device=dev)
        evs[-1].phase_hamiltonian(aob, buf)
        send.append(buf)
    recv = torch.stack(send)                                   # (world, G, chunk) = all_gather_into_tensor
    rows_all = recv.permute(1, 0, 2).reshape(G, world * chunk).contiguous()
    total = torch.zeros_like(evs[0].grad)
    for r, ev in enumerate(evs):
        ev.phase_solve(aob, rows_all, 1)
        ev.phase_gradient(aob, partial_rank=(r != 0))
        total += ev.grad
    torch.cuda.synchronize()
    for ev in evs:
        np.testing.assert_allclose(ev.energy[:, 0].cpu().numpy(), Eref, rtol=0, atol=1e-11)
    np.testing.assert_allclose(total.cpu().numpy(), gref, rtol=0, atol=1e-10)
