"""A few hosted H30 steps, to be run under rocprofv3 --kernel-trace --memory-copy-trace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from evcont_amd.evaluator import DeviceTRDMs
from evcont_amd.hosted import HostedEvaluator
from evcont_amd.synthetic import make_device_ao, make_device_trdm_rows
dev = torch.device("cuda:0")
n, A, T = 30, 30, 20
S, one, rows = make_device_trdm_rows(n, T, 2, 1236, dev)
trd = DeviceTRDMs.from_device_rows(one, rows, S, 2).compress_sym8_()
src = make_device_ao(n, A, 5, dev, ip1_rs_symmetric=True).packed_ip1(eri=True)
hv = HostedEvaluator(trd, A, src.aoslices.cpu().numpy(), warm_start=False, use_graph=False)
st = hv.staging()
for name in ("S", "hcore", "ipovlp", "dhcore", "gnuc", "eri", "eri_ip1"):
    np.copyto(st[name], getattr(src, name).cpu().numpy().reshape(st[name].shape))
for _ in range(8):
    hv.run()
