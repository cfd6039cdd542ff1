"""H2D bandwidth from pinned memory and the cost of the hosted step variants (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
dev = torch.device("cuda:0")
for mb in (0.1, 1, 12, 64):
    n = int(mb * 1e6 / 8)
    h = torch.zeros(n, dtype=torch.float64).pin_memory(); d = torch.zeros(n, dtype=torch.float64, device=dev)
    for _ in range(3): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"H2D {mb} MB pinned: {dt*1e6:.0f} us  {mb*1e6/dt/1e9:.1f} GB/s")
from evcont_amd.evaluator import DeviceTRDMs
from evcont_amd.hosted import HostedEvaluator
from evcont_amd.synthetic import make_device_ao, make_device_trdm_rows
for n, A, T in ((30, 30, 20), (10, 10, 5), (13, 3, 10)):
    print("workload", n, A, T)
    S, one, rows = make_device_trdm_rows(n, T, 2, 1236, dev)
    trd = DeviceTRDMs.from_device_rows(one, rows, S, 2).compress_sym8_()
    src = make_device_ao(n, A, 5, dev, ip1_rs_symmetric=True).packed_ip1(eri=True)
    for graph in (False, True):
        hv = HostedEvaluator(trd, A, src.aoslices.cpu().numpy(), warm_start=False, use_graph=graph)
        st = hv.staging()
        for name in ("S", "hcore", "ipovlp", "dhcore", "gnuc", "eri", "eri_ip1"):
            np.copyto(st[name], getattr(src, name).cpu().numpy().reshape(st[name].shape))
        for _ in range(4): hv.run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): hv.run()
        dt = (time.perf_counter() - t0) / 50
        print(f"hosted step graph={graph}: {dt*1e6:.0f} us")
        # device part only: same graph without copies is not available; time enqueue+phase on resident data
        t0 = time.perf_counter()
        for _ in range(50):
            hv.ev.enqueue(hv.aob, 1, energy_only=True); hv.ev.phase_gradient(hv.aob, False)
        hv.stream.synchronize(); dt = (time.perf_counter() - t0) / 50
        print(f"  device work only (eager, resident): {dt*1e6:.0f} us")
