"""Hosted MD step (integrals in pinned host memory) against the resident one, with and without copies:
python tools/micro/hosted_probe.py [H10|H2O|Zundel|H30 ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from evcont_amd.evaluator import DeviceTRDMs, ContinuationEvaluator
from evcont_amd.hosted import HostedEvaluator
from evcont_amd.synthetic import make_device_ao, make_device_trdm_rows
W = {"H30": (30, 30, 20, None), "H10": (10, 10, 5, None), "H2O": (13, 3, 10, (9, 2, 2)),
     "Zundel": (28, 7, 30, (9, 2, 2, 2, 9, 2, 2))}
dev = torch.device("cuda:0")
for name in sys.argv[1:] or ["H10", "H2O", "Zundel", "H30"]:
    n, A, T, sizes = W[name]
    S, one, rows = make_device_trdm_rows(n, T, 2, 1236, dev)
    trd = DeviceTRDMs.from_device_rows(one, rows, S, 2).compress_sym8_()
    srcs = [make_device_ao(n, A, 5 + k, dev, sizes, ip1_rs_symmetric=True).packed_ip1(eri=True) for k in range(4)]
    ev = ContinuationEvaluator(trd, A, want_two_rdm=False)
    for k in range(8): ev.enqueue(srcs[k % 4])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(100):
        ev.enqueue(srcs[k % 4]); ev.synchronize()
    res = (time.perf_counter() - t0) / 100
    out = [f"{name}: resident (sync every step) {res*1e6:.0f} us"]
    for zc in (False, True):
        hvs = []
        for k in range(4):
            hv = HostedEvaluator(trd, A, srcs[k].aoslices.cpu().numpy(), warm_start=False, zero_copy=zc)
            st = hv.staging()
            for nm in ("S", "hcore", "ipovlp", "dhcore", "gnuc", "eri", "eri_ip1"):
                np.copyto(st[nm], getattr(srcs[k], nm).cpu().numpy().reshape(st[nm].shape))
            st["enuc"][0] = srcs[k].enuc
            for _ in range(3): e, g = hv.run()
            hvs.append(hv)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(100): e, g = hvs[k % 4].run()
        dt = (time.perf_counter() - t0) / 100
        ev.enqueue(srcs[3]); ev.synchronize()
        de = abs(e - float(ev.energy[0].item())); dg = float(np.abs(g - ev.grad[:A].cpu().numpy()).max())
        out.append(f"hosted zero_copy={zc}: {dt*1e6:.0f} us (dE {de:.1e}, dgrad {dg:.1e})")
    print("; ".join(out), flush=True)
