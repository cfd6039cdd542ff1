"""Child process of tests/test_gpu_rccl.py: ONE rank, backend "nccl" (= RCCL on ROCm), the real HIP phases of the
pair-sharded evaluation (evcont_amd/distributed.py: phase A -> all_gather_into_tensor of the H rows -> phase B ->
phase C -> all_reduce of the gradient) against the CPU oracle (get_energy_with_grad,
ab_initio_gradients_loewdin.py:308-379).  Prints one JSON line with the worst differences."""
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
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAO, DeviceAOBatch, BatchedEvaluator, ContinuationEvaluator
    from evcont_amd.distributed import PairShardedContinuation, PipelinedPairSharded
    from oracle import evcont_oracle as orc

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)       # RANK / WORLD_SIZE / MASTER_* from the environment
    assert dist.get_world_size() == 1 and dist.get_backend() == "nccl"
    out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    n, T, A, G = 13, 9, 3, 17
    S, one, two = make_trdms(n, T, 4201)
    two_p = pack_rows(two, True, True)
    aos = [make_ao_arrays(n, A, 4300 + k, ao_sizes=(9, 2, 2), ip1_rs_symmetric=True) for k in range(G)]
    bundle = lambda ao: orc.AOBundle(ao.S, ao.hcore, ao.eri, ao.ipovlp, ao.dhcore, ao.eri_ip1, ao.aoslices, ao.enuc,
                                     ao.gnuc)
    want = [orc.energy_with_grad(bundle(ao), one, two_p, S) for ao in aos]
    rows = T * (T + 1) // 2
    worst_e, worst_g = 0.0, 0.0
    for comp in (None, "sym8"):
        packed = comp is not None
        trd = DeviceTRDMs(one, two_p, S, dev, compress=comp)
        # batched phases (packed s4 / s2kl inputs on the compressed layout: flags of ABI 7), two calls in a row
        psc = PairShardedContinuation(BatchedEvaluator(trd, A, G), rows)
        aob = DeviceAOBatch.from_arrays(aos, dev, pack_ip1=packed, pack_eri=packed)
        for _ in range(2):
            E, grad = psc.energy_with_grad(aob)
        worst_e = max(worst_e, max(abs(E[k] - want[k][0]) for k in range(G)))
        worst_g = max(worst_g, max(float(np.abs(grad[k] - want[k][1]).max()) for k in range(G)))
        # single-geometry phases
        ps1 = PairShardedContinuation(ContinuationEvaluator(trd, A), rows)
        E1, g1 = ps1.energy_with_grad(DeviceAO.from_arrays(aos[3], dev, pack_ip1=packed, pack_eri=packed))
        worst_e = max(worst_e, abs(E1 - want[3][0]))
        worst_g = max(worst_g, float(np.abs(g1 - want[3][1]).max()))
        # several batches in flight on private streams, collectives issued in program order
        runners = [PairShardedContinuation(BatchedEvaluator(trd, A, G, stream=torch.cuda.Stream(dev)), rows)
                   for _ in range(3)]
        for k in range(6):
            runners[k % 3].enqueue(aob)
        torch.cuda.synchronize(dev)
        for r in runners:
            Ek = r.ev.energy[:, 0].cpu().numpy()
            gk = r.ev.grad.cpu().numpy()
            worst_e = max(worst_e, max(abs(Ek[k] - want[k][0]) for k in range(G)))
            worst_g = max(worst_g, max(float(np.abs(gk[k] - want[k][1]).max()) for k in range(G)))
        # the same through the library's own pipeline: one caller stream, three batches in flight
        pp = PipelinedPairSharded(trd, A, G, rows)
        tickets = [pp.enqueue(aob) for _ in range(3)]
        for k in range(4):
            ev = pp.results(tickets[k])
            Ek, gk = ev.energy[:, 0].clone(), ev.grad.clone()
            tickets.append(pp.enqueue(aob))
            torch.cuda.current_stream(dev).synchronize()
            Ek, gk = Ek.cpu().numpy(), gk.cpu().numpy()
            worst_e = max(worst_e, max(abs(Ek[j] - want[j][0]) for j in range(G)))
            worst_g = max(worst_g, max(float(np.abs(gk[j] - want[j][1]).max()) for j in range(G)))
        pp.synchronize()
    out.update(worst_dE=worst_e, worst_dgrad=worst_g)
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_CHILD " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
