import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from test_gpu_warm_start import blend
from evcont_amd.synthetic import make_ao_arrays, make_trdms, pack_rows
from evcont_amd.evaluator import DeviceTRDMs, DeviceAO, ContinuationEvaluator
from oracle import evcont_oracle as orc
dev = torch.device("cuda:0")
n, T, A = 13, 5, 3
S, one, two = make_trdms(n, T, 70 + n)
two_l = pack_rows(two, True, True)
trd = DeviceTRDMs(one, two_l, S, dev)
cold = ContinuationEvaluator(trd, A); warm = ContinuationEvaluator(trd, A, warm_start=True)
a0, a1, other = make_ao_arrays(n, A, 1), make_ao_arrays(n, A, 2), make_ao_arrays(n, A, 3)
seq = [blend(a0, a1, 0.002 * k) for k in range(5)] + [other, blend(a0, a1, 0.01)]
for k, ao in enumerate(seq):
    dao = DeviceAO.from_arrays(ao, dev)
    Ec, gc, Dc, Gc = cold.energy_with_grad(dao, True)
    Ew, gw, Dw, Gw = warm.energy_with_grad(dao, True)
    b = orc.AOBundle(ao.S, ao.hcore, ao.eri, ao.ipovlp, ao.dhcore, ao.eri_ip1, ao.aoslices, ao.enuc, ao.gnuc)
    Eo, go = orc.energy_with_grad(b, one, two_l, S)
    print(k, "E c-w %.1e c-o %.1e w-o %.1e | g c-w %.1e c-o %.1e w-o %.1e | D %.1e" % (abs(Ec-Ew), abs(Ec-Eo), abs(Ew-Eo), np.abs(gc-gw).max(), np.abs(gc-go).max(), np.abs(gw-go).max(), np.abs(Dc-Dw).max()))
    ew, cw = warm.energies(dao, nroots=3)
