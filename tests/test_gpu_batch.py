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


@pytest.mark.parametrize("lname", ["pack2", "full6", "sym8"])
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
    # "sym8": the 8-fold compressed device layout built from (shards of) the pack2 rows; integrals with the
    # symmetries of real ones, as that layout requires
    comp = "sym8" if lname == "sym8" else None
    two_l = layout(two, "pack2" if comp else lname)
    rows, _ = layout_shape(two_l.ndim, T, n)
    aob = DeviceAOBatch.stack([DeviceAO.from_arrays(make_ao_arrays(n, A, 900 + k, ip1_rs_symmetric=bool(comp)), dev)
                               for k in range(G)])
    full = BatchedEvaluator(DeviceTRDMs(one, two_l, S, dev, compress=comp), A, G)
    Eref, gref = full.energies_with_grads(aob)
    chunk = -(-rows // world)
    evs, send = [], []
    for r in range(world):
        r0, r1 = shard_rows(rows, world, r)
        evs.append(BatchedEvaluator(DeviceTRDMs(one, two_l, S, dev, row_range=(r0, r1), compress=comp), A, G))
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


@pytest.mark.parametrize("n,T,A,G,lname", [
    (18, 6, 3, 9, "pack2"),      # 16 < N <= 32: pair transform <32> with a ragged last q tile; 9 = 8 + 1 geometries
    (21, 5, 3, 17, "pack2"),     # odd N, two matrix-core groups (16 + 1)
    (18, 4, 2, 13, "full6"),     # general (unpacked) path, one matrix-core group of 13
])
def test_batch_vs_oracle_midsize(n, T, A, G, lname):
    """Batched pipeline at sizes where the batch-only code paths are live (matrix-core streaming kernels,
    XCD-ordered unpack with a remainder, several q tiles per workgroup), against the CPU oracle."""
    from oracle import evcont_oracle as orc
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAO, DeviceAOBatch, BatchedEvaluator
    dev = torch.device("cuda:0")
    S, one, two = make_trdms(n, T, 300 + n)
    two_l = layout(two, lname)
    aos = [make_ao_arrays(n, A, 800 + k) for k in range(G)]
    be = BatchedEvaluator(DeviceTRDMs(one, two_l, S, dev), A, G, keep_density_matrices=True)
    E, grad = be.energies_with_grads(DeviceAOBatch.stack([DeviceAO.from_arrays(a, dev) for a in aos]))
    for k in (0, 7, 8, G - 1):
        a = aos[k]
        b = orc.AOBundle(a.S, a.hcore, a.eri, a.ipovlp, a.dhcore, a.eri_ip1, a.aoslices, a.enuc, a.gnuc)
        Eo, go, Do, Go = orc.energy_with_grad(b, one, two_l, S, True, True)
        assert abs(E[k] - Eo) < 1e-9, k
        np.testing.assert_allclose(grad[k], go, rtol=0, atol=1e-8)
        np.testing.assert_allclose(be.d_pred[k].cpu().numpy(), Do, rtol=0, atol=1e-10)
        np.testing.assert_allclose(be.g_pred[k].cpu().numpy(), Go, rtol=0, atol=1e-10)


def test_h30_batch16_matches_single_full_size():
    """BASELINE metric configuration at FULL size (N=30, A=30, T=20, packed 210 x 405450 t-RDM, 16 geometries
    per pass): the batched pipeline must reproduce the single-geometry pipeline, which is pinned to the
    reference at small sizes and by the layout-equivalence property at N=30."""
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAOBatch, ContinuationEvaluator, BatchedEvaluator
    from evcont_amd.synthetic import make_device_ao, make_device_trdm_rows
    dev = torch.device("cuda:0")
    n, A, T, G = 30, 30, 20, 16
    S, one, rows = make_device_trdm_rows(n, T, 2, 1236, dev)
    trd = DeviceTRDMs.from_device_rows(one, rows, S, 2)
    del rows
    aos = [make_device_ao(n, A, 5000 + k, dev) for k in range(G)]
    be = BatchedEvaluator(trd, A, G)
    be.enqueue(DeviceAOBatch.stack(aos))
    be.synchronize()
    E = be.energy[:, 0].cpu().numpy()
    grad = be.grad.cpu().numpy()
    assert np.all(np.isfinite(E)) and np.all(np.isfinite(grad))
    single = ContinuationEvaluator(trd, A)
    for k in (0, 5, 15):
        Es, gs = single.energy_with_grad(aos[k])
        assert abs(E[k] - Es) < 1e-9, (k, E[k], Es)
        np.testing.assert_allclose(grad[k], gs, rtol=0, atol=1e-8)
    # a second, identical pass is bit-identical (fixed reduction orders, no atomics)
    be.enqueue(DeviceAOBatch.stack(aos))
    be.synchronize()
    assert np.array_equal(be.energy[:, 0].cpu().numpy(), E) and np.array_equal(be.grad.cpu().numpy(), grad)


def test_phase_loewdin_then_flagged_call_equals_fused():
    """evc_phase_loewdin_batch + EVC_FLAG_LOEWDIN_DONE (the Loewdin kernel of a batch run ahead of time, e.g. on
    another stream beside the previous batch) gives the results of the fused call -- to rounding: a fused call of fewer
    than 12 geometries takes X = S^-1/2 from the Newton-Schulz iteration, the phase from the eigendecomposition."""
    import torch
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAOBatch, BatchedEvaluator
    from evcont_amd.synthetic import make_ao_arrays, make_trdms, pack_rows
    dev = torch.device("cuda:0")
    n, T, A, G = 13, 5, 3, 5
    S, one, two = make_trdms(n, T, 61)
    trd = DeviceTRDMs(one, pack_rows(two, True, True), S, dev, compress="sym8")
    aob = DeviceAOBatch.from_arrays([make_ao_arrays(n, A, 900 + k, ao_sizes=(9, 2, 2), ip1_rs_symmetric=True)
                                     for k in range(G)], dev, pack_ip1=True, pack_eri=True)
    ref = BatchedEvaluator(trd, A, G)
    E0, g0 = ref.energies_with_grads(aob)
    be = BatchedEvaluator(trd, A, G)
    side = torch.cuda.Stream(dev)
    be.phase_loewdin(aob, stream=side)
    side.synchronize()
    E1, g1 = be.energies_with_grads(aob)
    np.testing.assert_allclose(E1, E0, rtol=0, atol=1e-12)
    np.testing.assert_allclose(g1, g0, rtol=0, atol=1e-11)
    E2, g2 = be.energies_with_grads(aob)          # the flag is consumed by one call
    assert np.array_equal(E0, E2) and np.array_equal(g0, g2)


def test_pipelined_batched_evaluator_matches_plain():
    """PipelinedBatchedEvaluator (one caller stream, `depth` batches in flight on the library's internal streams):
    same numbers as the plain evaluator for a sequence of different batches, WITHOUT a host synchronisation between
    submissions; results are read on the caller's stream behind ``results(ticket)`` only.  The inputs of the last
    batches are produced asynchronously on the caller's stream right before their submission (a batch must not start
    before what its caller had enqueued)."""
    import torch
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAOBatch, BatchedEvaluator, PipelinedBatchedEvaluator
    from evcont_amd.synthetic import make_ao_arrays, make_trdms, pack_rows
    dev = torch.device("cuda:0")
    n, T, A, G = 13, 5, 3, 4
    S, one, two = make_trdms(n, T, 62)
    trd = DeviceTRDMs(one, pack_rows(two, True, True), S, dev, compress="sym8")
    batches = [DeviceAOBatch.from_arrays([make_ao_arrays(n, A, 950 + 10 * b + k, ao_sizes=(9, 2, 2), ip1_rs_symmetric=True)
                                          for k in range(G)], dev, pack_ip1=True, pack_eri=True) for b in range(5)]
    ref = BatchedEvaluator(trd, A, G)
    want = [ref.energies_with_grads(b) for b in batches]
    for depth in (1, 2, 3):
        pe = PipelinedBatchedEvaluator(trd, A, G, depth=depth)
        order = [0, 1, 2, 3, 4, 2, 0, 4, 1]
        got, tickets, keep = {}, [], []      # keep: inputs stay allocated while their batch is in flight
        caller = torch.cuda.Stream(dev)
        with torch.cuda.stream(caller):
            for i, b in enumerate(order):
                if i >= depth:                     # consume the batch whose slot is about to be reused
                    r = pe.results(tickets[i - depth])
                    got[i - depth] = (r.energy[:, 0].clone(), r.grad[:, :A].clone())
                src = batches[b]
                if i >= 5:                         # inputs written by kernels still queued on the caller's stream
                    src = DeviceAOBatch(S=torch.empty_like(src.S), hcore=src.hcore, eri=torch.empty_like(src.eri),
                                        enuc=src.enuc, natm=src.natm, ipovlp=src.ipovlp, dhcore=src.dhcore,
                                        eri_ip1=src.eri_ip1, gnuc=src.gnuc, aoslices=src.aoslices,
                                        ip1_s2kl=src.ip1_s2kl, eri_s4=src.eri_s4)
                    big = torch.randn(1 << 24, device=dev)          # something that takes a while in front of the copies
                    big = (big * big).sum()
                    src.S.copy_(batches[b].S, non_blocking=True)
                    src.eri.copy_(batches[b].eri, non_blocking=True)
                keep.append(src)
                tickets.append(pe.enqueue(src))
            for i in range(len(order) - depth, len(order)):
                r = pe.results(tickets[i])
                got[i] = (r.energy[:, 0].clone(), r.grad[:, :A].clone())
        caller.synchronize()
        for i, b in enumerate(order):
            assert np.array_equal(got[i][0].cpu().numpy(), want[b][0]), (depth, i)
            assert np.array_equal(got[i][1].cpu().numpy(), want[b][1]), (depth, i)
