#!/usr/bin/env python3
"""Throughput benchmark of the continuation energy+force hot path (BASELINE.json metric:
continuation geometries/sec, energy+force, H30 STO-3G, 20 training states).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--layout pack2|full6|pair5|elec3]

One "step" = one energy+force evaluation of one synthetic geometry (N=30 orbitals, A=30
atoms) against T=20 training states whose t-RDMs are resident in HBM.  64 distinct geometry
bundles are resident on the device and cycled.  With --gpus N > 1 (launched by
torch.distributed.run, one rank per GPU) the training PAIRS are sharded over the ranks and
each evaluation uses two KB-sized RCCL collectives (evcont_amd/distributed.py): total work is
fixed, so scaling is "strong".

Rank 0 prints ONE JSON line (see DESIGN.md §Measurement for every field).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np
import torch
import torch.distributed as dist

WORKLOADS = {
    # name: (N, A, T, ao_sizes)
    "H30": (30, 30, 20, None),            # BASELINE configs[2], the metric's configuration
    "H10": (10, 10, 5, None),             # configs[1]
    "H2O": (13, 3, 10, (9, 2, 2)),        # configs[3] (T assumed 10)
    "Zundel": (28, 7, 30, (9, 2, 2, 2, 9, 2, 2)),  # configs[4]
}
LAYOUT_ND = {"full6": 6, "pair5": 5, "elec3": 3, "pack2": 2}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable by a float4 copy)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--workload", default="H30", choices=list(WORKLOADS))
    p.add_argument("--layout", default="pack2", choices=list(LAYOUT_ND))
    p.add_argument("--geoms", type=int, default=64, help="distinct synthetic geometries resident on the device")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-samples", type=int, default=0, help="geometries timed on the host (0 = auto)")
    p.add_argument("--energy-only", action="store_true")
    p.add_argument("--batch", type=int, default=1,
                   help="geometries per step: every launch covers the batch and the t-RDM is streamed once per "
                        "8 geometries (evc_energy_with_grad_batch); 1 = one geometry per step as in an MD run")
    p.add_argument("--streams", type=int, default=1,
                   help="independent geometries in flight (one HIP stream + workspace each); 1 = strictly "
                        "sequential evaluations as in an MD run")
    return p.parse_args()


def cpu_baseline(workload, layout_nd, trd, aos, samples):
    """Time the numpy oracle (a restatement of the reference's CPU algorithm, kind="port") on the
    host cores for a bounded sample of the same workload."""
    from oracle import evcont_oracle as orc
    n, A, T, _ = WORKLOADS[workload]
    two = trd.two[: trd.rows_local, : trd.cols].cpu().numpy()
    if layout_nd == 6:
        two = two.reshape(T, T, n, n, n, n)
    elif layout_nd == 5:
        two = two.reshape(-1, n, n, n, n)
    elif layout_nd == 3:
        two = two.reshape(T, T, -1)
    one = trd.one[:, : n * n].cpu().numpy().reshape(T, T, n, n)
    S = trd.S.cpu().numpy()
    times = []
    for k in range(samples + 1):
        ao = aos[k % len(aos)]
        b = orc.AOBundle(S=ao.S.cpu().numpy(), hcore=ao.hcore.cpu().numpy(), eri=ao.eri.cpu().numpy(),
                         ipovlp=ao.ipovlp.cpu().numpy(), dhcore=ao.dhcore.cpu().numpy(),
                         eri_ip1=ao.eri_ip1.cpu().numpy(), aoslices=ao.aoslices.cpu().numpy(),
                         enuc=ao.enuc, gnuc=ao.gnuc.cpu().numpy())
        t0 = time.perf_counter()
        orc.energy_with_grad(b, one, two, S)
        dt = time.perf_counter() - t0
        if k > 0:           # first call = warm-up (BLAS thread pool, page faults)
            times.append(dt)
    med = float(np.median(times))
    try:
        from threadpoolctl import threadpool_info
        cores = max([d.get("num_threads", 1) for d in threadpool_info()] or [os.cpu_count() or 1])
    except Exception:
        cores = os.cpu_count() or 1
    return {"value": 1.0 / med, "unit": "geometries/s", "cores": int(cores), "kind": "port",
            "sample": f"{samples} energy+force evaluations of the same {workload} workload "
                      f"(layout ndim {layout_nd}) by oracle/evcont_oracle.py (numpy/OpenBLAS), median, after 1 warm-up"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from evcont_amd import _lib
    from evcont_amd.evaluator import (DeviceTRDMs, ContinuationEvaluator, BatchedEvaluator, DeviceAOBatch,
                                      layout_shape)
    from evcont_amd.distributed import PairShardedContinuation, shard_rows
    from evcont_amd.synthetic import make_device_ao, make_device_trdm_rows

    n, A, T, sizes = WORKLOADS[a.workload]
    nd = LAYOUT_ND[a.layout]
    rows, cols = layout_shape(nd, T, n)
    seed = 1234 + list(WORKLOADS).index(a.workload)
    r0, r1 = shard_rows(rows, world, rank)
    S_train, one, two_rows = make_device_trdm_rows(n, T, nd, seed, dev, (r0, r1))
    trd = DeviceTRDMs.from_device_rows(one, two_rows, S_train, nd, r0, rows)
    del two_rows
    aos = [make_device_ao(n, A, seed * 1000 + k, dev, sizes) for k in range(a.geoms)]
    nslots = max(1, a.streams) if world == 1 else 1
    G = max(1, a.batch) if world == 1 else 1
    mk_stream = lambda: (torch.cuda.Stream(dev) if nslots > 1 else None)
    if G > 1:
        nb = max(1, len(aos) // G)
        batches = [DeviceAOBatch.stack(aos[i * G:(i + 1) * G]) for i in range(nb)]
        evs = [BatchedEvaluator(trd, A, G, stream=mk_stream()) for _ in range(nslots)]
    else:
        evs = [ContinuationEvaluator(trd, A, stream=mk_stream()) for _ in range(nslots)]
    ev = evs[0]
    runner = PairShardedContinuation(ev, rows) if world > 1 else None
    lib = _lib.load()

    def step(k):
        if runner is not None:
            runner.enqueue(aos[k % len(aos)], 1, a.energy_only)
        elif G > 1:
            evs[k % nslots].enqueue(batches[k % len(batches)], 1, a.energy_only)
        else:
            evs[k % nslots].enqueue(aos[k % len(aos)], 1, a.energy_only)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for k in range(a.warmup):
        step(k)
    fence()
    _lib.check(lib.evc_profile_begin(a.steps), "evc_profile_begin")
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(a.warmup + k)
    fence()
    dt = time.perf_counter() - t0
    rows_ms, cols_ms = C.c_double(), C.c_double()
    rows_n, cols_n = C.c_int(), C.c_int()
    _lib.check(lib.evc_profile_end(C.byref(rows_ms), C.byref(rows_n), C.byref(cols_ms), C.byref(cols_n)),
               "evc_profile_end")
    e_last = float(ev.energy.reshape(-1)[0].item())
    assert np.isfinite(e_last), "non-finite energy in the timed region"
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        # dominant kernel = K5, the 2-RDM x ERI contraction (rows GEMV); algorithmic bytes per
        # launch = local two-body rows x cols x 8 B + one-body rows + the two vectors
        # (+ per geometry of the batch: the h2 / predicted-RDM vector).  A batch of G > 8 is G/8 launches.
        gl = min(G, 8)                      # geometries per launch
        launches_per_step = -(-G // 8)
        bytes_rows = trd.rows_local * cols * 8 + T * T * n * n * 8 + gl * (cols * 8 + n * n * 8)
        bytes_cols = bytes_rows
        k5_ms = rows_ms.value / max(rows_n.value, 1) / launches_per_step
        k8_ms = cols_ms.value / max(cols_n.value, 1) / launches_per_step if cols_n.value else None
        ach = bytes_rows / (k5_ms * 1e-3) / 1e9
        traffic = None
        tj = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(tj):
            try:
                rec = json.load(open(tj))
                key = f"{a.workload}/{a.layout}/gemv_rows"
                if key in rec:
                    traffic = rec[key]["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        out = {
            "metric": "continuation geometries/sec (energy+force), H30 STO-3G, 20 training states"
            if a.workload == "H30" and not a.energy_only else
            f"continuation geometries/sec ({'energy' if a.energy_only else 'energy+force'}), {a.workload}",
            "value": a.steps * G / dt,
            "unit": "geometries/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{a.workload}: N={n} orbitals, A={A} atoms, T={T} training states, "
                                   f"two-body t-RDM layout {a.layout} ({rows}x{cols} f64, "
                                   f"{rows * cols * 8 / 1e9:.3f} GB resident in HBM), {a.geoms} resident geometries",
                       "parallelism": f"pairs{world}" if world > 1 else "single",
                       "streams": nslots, "geometries_per_step": G},
            "roofline": {"bound": "hbm", "kernel": "gemv_rows_kernel (K5: H_ab = Gamma.h2)",
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": traffic, "bytes_per_launch": bytes_rows, "ms_per_launch": k5_ms,
                         "launches": rows_n.value * launches_per_step, "geometries_per_launch": gl},
            "kernels": {"gemv_rows_ms": k5_ms, "gemv_cols_ms": k8_ms,
                        "gemv_cols_GBs": (bytes_cols / (k8_ms * 1e-3) / 1e9) if k8_ms else None},
            "last_energy": e_last,
        }
        if not a.no_cpu_baseline and world == 1 and not a.energy_only:
            samples = a.cpu_samples or (8 if a.workload in ("H30", "Zundel") else 50)
            out["cpu_baseline"] = cpu_baseline(a.workload, nd, trd, aos, samples)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
