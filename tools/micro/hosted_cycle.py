import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from evcont_amd.evaluator import DeviceTRDMs
from evcont_amd.hosted import HostedEvaluator
from evcont_amd.synthetic import make_device_ao, make_device_trdm_rows
dev = torch.device("cuda:0")
n, A, T = 30, 30, 20
S, one, rows = make_device_trdm_rows(n, T, 2, 1236, dev)
trd = DeviceTRDMs.from_device_rows(one, rows, S, 2).compress_sym8_()
hvs = []
for k in range(4):
    src = make_device_ao(n, A, 5 + k, dev, ip1_rs_symmetric=True).packed_ip1(eri=True)
    hv = HostedEvaluator(trd, A, src.aoslices.cpu().numpy(), warm_start=False)
    st = hv.staging()
    for name in ("S", "hcore", "ipovlp", "dhcore", "gnuc", "eri", "eri_ip1"):
        np.copyto(st[name], getattr(src, name).cpu().numpy().reshape(st[name].shape))
    for _ in range(3): hv.run()
    hvs.append(hv)
for nev in (1, 2, 4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(60): hvs[k % nev].run()
    dt = (time.perf_counter() - t0) / 60
    print(f"{nev} evaluator(s) cycled: {dt*1e6:.0f} us per step")
