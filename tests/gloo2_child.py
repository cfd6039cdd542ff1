"""Child process of tests/test_gpu_rccl.py::test_pipelined_pair_sharded_two_ranks: ONE of TWO ranks that share the card,
backend "gloo" (the only way to have more than one rank on a one-GPU box; the collectives take the device tensors): each
rank holds its row slice of the compressed training set and drives the real HIP phases through
``distributed.PipelinedPairSharded`` (three pair-sharded batches in flight on internal streams, the collectives of all of
them issued in program order on one communicator) and through ``PairShardedContinuation(return_density_matrices=True)``,
against the CPU oracle.  Rank 0 prints one JSON line."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np
import torch
import torch.distributed as dist


def main():
    from evcont_amd.synthetic import make_ao_arrays, make_trdms, pack_rows
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAOBatch, BatchedEvaluator
    from evcont_amd.distributed import PairShardedContinuation, PipelinedPairSharded, shard_rows
    from oracle import evcont_oracle as orc

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo")
    world, rank = dist.get_world_size(), dist.get_rank()
    n, T, A, G = 13, 7, 3, 14
    S, one, two = make_trdms(n, T, 4601)
    two_p = pack_rows(two, True, True)
    batches = [[make_ao_arrays(n, A, 4700 + 50 * b + k, ao_sizes=(9, 2, 2), ip1_rs_symmetric=True) for k in range(G)]
               for b in range(4)]
    bundle = lambda ao: orc.AOBundle(ao.S, ao.hcore, ao.eri, ao.ipovlp, ao.dhcore, ao.eri_ip1, ao.aoslices, ao.enuc,
                                     ao.gnuc)
    rows = T * (T + 1) // 2
    r0, r1 = shard_rows(rows, world, rank)
    trd = DeviceTRDMs(one, two_p, S, dev, row_range=(r0, r1), compress="sym8")
    aobs = [DeviceAOBatch.from_arrays(b, dev, pack_ip1=True, pack_eri=True) for b in batches]
    worst_e = worst_g = worst_d = 0.0
    # one caller stream, three pair-sharded batches in flight inside the library; a different batch in every slot
    pp = PipelinedPairSharded(trd, A, G, rows)
    order = [0, 1, 2, 3, 1, 0]
    tickets = [pp.enqueue(aobs[b]) for b in order[:3]]
    for k, b in enumerate(order):
        ev = pp.results(tickets[k])
        Ek, gk = ev.energy[:, 0].clone(), ev.grad.clone()
        if k + 3 < len(order):
            tickets.append(pp.enqueue(aobs[order[k + 3]]))
        torch.cuda.current_stream(dev).synchronize()
        Ek, gk = Ek.cpu().numpy(), gk.cpu().numpy()
        for j in (0, G // 2, G - 1):
            Eo, go = orc.energy_with_grad(bundle(batches[b][j]), one, two_p, S)
            worst_e = max(worst_e, abs(Ek[j] - Eo))
            worst_g = max(worst_g, float(np.abs(gk[j][:A] - go).max()))
    pp.synchronize()
    # the plain runner with the predicted RDMs summed over the ranks
    psc = PairShardedContinuation(BatchedEvaluator(trd, A, G, keep_density_matrices=True), rows, return_density_matrices=True)
    E, grad, D, Gm = psc.energy_with_grad(aobs[2], True)
    # ... against the complete training set on one device (same kernels, no sharding) and the oracle
    full = BatchedEvaluator(DeviceTRDMs(one, two_p, S, dev, compress="sym8"), A, G, keep_density_matrices=True)
    Ef, gf = full.energies_with_grads(aobs[2])
    Df, Gf = full.d_pred.cpu().numpy(), full.g_pred.cpu().numpy()
    worst_d = max(float(np.abs(D - Df).max()), float(np.abs(Gm - Gf).max()))
    worst_e = max(worst_e, float(np.abs(E - Ef).max()))
    worst_g = max(worst_g, float(np.abs(grad - gf).max()))
    for j in (0, G - 1):
        Eo, go = orc.energy_with_grad(bundle(batches[2][j]), one, two_p, S)
        worst_e = max(worst_e, abs(E[j] - Eo))
        worst_g = max(worst_g, float(np.abs(grad[j][:A] - go).max()))
    dist.barrier()
    if rank == 0:
        print("GLOO2_CHILD " + json.dumps({"world": world, "rows": [r0, r1], "worst_dE": worst_e, "worst_dgrad": worst_g,
                                          "worst_drdm": worst_d}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
