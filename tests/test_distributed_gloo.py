"""World-size-2/3 CPU (gloo) tests of the pair-sharded driver (evcont_amd/distributed.py).
The three device phases are replaced by a test double built on the CPU oracle, so what is
tested here is the sharding, the two collectives and the linear split of the gradient; the
HIP phases themselves are covered by the -m gpu parity tests."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from evcont_amd.synthetic import make_ao_arrays, make_trdms, pack_rows
from evcont_amd.distributed import PairShardedContinuation, shard_rows
from oracle import evcont_oracle as orc


class OraclePhases:
    """CPU stand-in for evaluator.ContinuationEvaluator restricted to rows [r0, r1) (pack2 layout)."""

    def __init__(self, one, two_packed_rows, S, r0, rows_total, natm):
        self.one, self.two, self.S = one, two_packed_rows, S
        self.r0, self.rows_total = r0, rows_total
        self.T, self.n = S.shape[0], one.shape[-1]
        self.grad = torch.zeros((natm, 3), dtype=torch.float64)
        self.energy = torch.zeros(self.T, dtype=torch.float64)
        self.d_pred = torch.zeros((self.n, self.n), dtype=torch.float64)
        self.g_pred = torch.zeros((self.n,) * 4, dtype=torch.float64)

    def _bundle(self, ao):
        return orc.AOBundle(ao.S, ao.hcore, ao.eri, ao.ipovlp, ao.dhcore, ao.eri_ip1, ao.aoslices, ao.enuc, ao.gnuc)

    def phase_hamiltonian(self, ao):
        b = self._bundle(ao)
        self.X = orc.loewdin_trafo(b.S)
        self.h1, self.h2 = orc.integrals_oao(b, self.X)
        return torch.from_numpy(self.two @ orc.pack_pair_sym(self.h2, 0.5))

    def phase_solve(self, ao, rows_all, nroots=1):
        H = np.tensordot(self.one, self.h1, axes=2)
        H[np.tril_indices(self.T)] += rows_all.numpy()
        e, c = orc._gen_eig(H, self.S, True)
        k = int(np.argmin(e.real))
        self.c = c[:, k].real
        self.energy[0] = float(e[k].real) + ao.enuc

    def phase_gradient(self, ao, partial_rank):
        b = self._bundle(ao)
        w = orc.pair_weights(self.c)[self.r0: self.r0 + self.two.shape[0]]
        G = orc.unpack_pair_sym(w @ self.two, self.n) if len(w) else np.zeros((self.n,) * 4)
        dX = orc.derivative_ao_mo_trafo(b)
        g = 0.5 * orc.two_el_grad(b.eri, G, self.X, dX, b.eri_ip1, [tuple(s) for s in b.aoslices])
        # as the device phases: the PARTIAL unpacked 2-RDM of the local rows, the complete 1-RDM on every rank
        self.g_pred.copy_(torch.from_numpy(np.ascontiguousarray(G)))
        self.d_pred.copy_(torch.from_numpy(np.tensordot(np.outer(self.c, self.c), self.one, axes=2)))
        if not partial_rank:
            D = np.tensordot(np.outer(self.c, self.c), self.one, axes=2)
            g = g + np.tensordot(D, orc.one_el_grad(b, self.X, dX), axes=([0, 1], [0, 1])) + b.gnuc
        self.grad.copy_(torch.from_numpy(g))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, T, A, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ao = make_ao_arrays(n, A, 5)
        S, one, two = make_trdms(n, T, 6)
        packed = pack_rows(two, True, True)
        rows = packed.shape[0]
        r0, r1 = shard_rows(rows, world, rank)
        ev = OraclePhases(one, packed[r0:r1], S, r0, rows, A)
        drv = PairShardedContinuation(ev, rows)
        assert (drv.r0, drv.r1) == (r0, r1)
        E, g = drv.energy_with_grad(ao)
        # a second geometry through the same driver (buffers are reused)
        ao2 = make_ao_arrays(n, A, 7)
        E2, g2 = drv.energy_with_grad(ao2)
        # the predicted RDMs (return_density_matrices of the reference): the 2-RDM is reduced over the ranks
        drv_r = PairShardedContinuation(OraclePhases(one, packed[r0:r1], S, r0, rows, A), rows,
                                        return_density_matrices=True)
        E3, g3, D3, G3 = drv_r.energy_with_grad(ao, return_density_matrices=True)
        assert abs(E3 - E) < 1e-13 and np.abs(g3 - g).max() < 1e-13
        q.put((rank, E, g, E2, g2, D3, G3))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,T", [(2, 3), (3, 4), (2, 1)])
def test_pair_sharded_matches_single(world, T):
    n, A = 4, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, T, A, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    S, one, two = make_trdms(n, T, 6)
    packed = pack_rows(two, True, True)
    for seed, (ie, ig) in ((5, (1, 2)), (7, (3, 4))):
        ao = make_ao_arrays(n, A, seed)
        b = orc.AOBundle(ao.S, ao.hcore, ao.eri, ao.ipovlp, ao.dhcore, ao.eri_ip1, ao.aoslices, ao.enuc, ao.gnuc)
        Eref, gref = orc.energy_with_grad(b, one, packed, S)
        for r in res:
            assert abs(r[ie] - Eref) < 1e-11              # every rank holds the same energy
            np.testing.assert_allclose(r[ig], gref, rtol=0, atol=1e-10)   # ... and the reduced gradient
    ao = make_ao_arrays(n, A, 5)
    b = orc.AOBundle(ao.S, ao.hcore, ao.eri, ao.ipovlp, ao.dhcore, ao.eri_ip1, ao.aoslices, ao.enuc, ao.gnuc)
    _, _, Dref, Gref = orc.energy_with_grad(b, one, packed, S, True, True)
    for r in res:                                          # every rank holds the complete predicted RDMs
        np.testing.assert_allclose(r[5], Dref, rtol=0, atol=1e-11)
        np.testing.assert_allclose(r[6], np.asarray(Gref).reshape(r[6].shape), rtol=0, atol=1e-11)


class OraclePhasesBatch:
    """Batched counterpart (evaluator.BatchedEvaluator phase API): `count` geometries per call, the rows go
    into the caller's (count, chunk) buffer."""

    def __init__(self, one, two_packed_rows, S, r0, rows_total, natm, count):
        self.count = count
        self.one_ev = [OraclePhases(one, two_packed_rows, S, r0, rows_total, natm) for _ in range(count)]
        self.grad = torch.zeros((count, natm, 3), dtype=torch.float64)
        self.energy = torch.zeros((count, S.shape[0]), dtype=torch.float64)

    def phase_hamiltonian(self, aos, rows_out):
        assert rows_out.shape[0] == self.count
        for g, (ev, ao) in enumerate(zip(self.one_ev, aos)):
            rows = ev.phase_hamiltonian(ao)
            rows_out[g, : rows.numel()].copy_(rows)

    def phase_solve(self, aos, rows_all, nroots=1):
        P = self.one_ev[0].rows_total
        for g, (ev, ao) in enumerate(zip(self.one_ev, aos)):
            ev.phase_solve(ao, rows_all[g, :P], nroots)
            self.energy[g] = ev.energy

    def phase_gradient(self, aos, partial_rank):
        for g, (ev, ao) in enumerate(zip(self.one_ev, aos)):
            ev.phase_gradient(ao, partial_rank)
            self.grad[g] = ev.grad


def _worker_batch(rank, world, port, n, T, A, seeds, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S, one, two = make_trdms(n, T, 6)
        packed = pack_rows(two, True, True)
        rows = packed.shape[0]
        r0, r1 = shard_rows(rows, world, rank)
        ev = OraclePhasesBatch(one, packed[r0:r1], S, r0, rows, A, len(seeds))
        drv = PairShardedContinuation(ev, rows)
        E, g = drv.energy_with_grad([make_ao_arrays(n, A, s) for s in seeds])
        q.put((rank, E, g))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,T", [(2, 3), (3, 2)])
def test_pair_sharded_batch_matches_single(world, T):
    """Batched driver: (G, chunk) send buffer, (world, G, chunk) gather and its re-ordering to pair order."""
    n, A, seeds = 4, 2, (5, 7, 9)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_batch, args=(r, world, port, n, T, A, seeds, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    S, one, two = make_trdms(n, T, 6)
    packed = pack_rows(two, True, True)
    for k, seed in enumerate(seeds):
        ao = make_ao_arrays(n, A, seed)
        b = orc.AOBundle(ao.S, ao.hcore, ao.eri, ao.ipovlp, ao.dhcore, ao.eri_ip1, ao.aoslices, ao.enuc, ao.gnuc)
        Eref, gref = orc.energy_with_grad(b, one, packed, S)
        for r in res:
            assert abs(r[1][k] - Eref) < 1e-11
            np.testing.assert_allclose(r[2][k], gref, rtol=0, atol=1e-10)
