"""Timing of the two single-workgroup eigen-kernels alone (one geometry): python tools/micro/loewdin_time.py [n] [T]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from evcont_amd import ops
from evcont_amd.synthetic import make_ao_arrays, make_trdms
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
ao = make_ao_arrays(n, 3, 1)
S = torch.from_numpy(ao.S).to(dev); h = torch.from_numpy(ao.hcore).to(dev)
def timeit(f, reps=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
import ctypes as C
from evcont_amd import _lib
lib = _lib.load()
def dbg(tag):
    if not hasattr(lib, "evc_debug_read"):      # product library: no stamps
        return
    st = (C.c_longlong * 64)(); va = (C.c_double * 64)()
    fn = lib.evc_debug_read; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    fn(st, va, 64)
    t = [ (st[i] - st[0]) / 100.0 for i in range(14)]
    print(tag, 'eigh: Af ready %.1f, start vectors %.1f (householder %.1f, multisection %.1f, vectors %.1f), refine passes %s, done %.1f' % (t[1], t[2], t[11], t[12], t[13], [round(t[i],1) for i in (3,4,5)], t[10]))
    print(tag, "pass 0: start %.1f, W done %.1f, S/R done %.1f, lam %.1f, elementwise %.1f, reduced %.1f" % tuple((st[i] - st[0]) / 100.0 for i in (14, 15, 16, 17, 18, 3)))
    print(tag, "kernel phases", [round((st[i]-st[30])/100.0,1) for i in range(30,38)], "few: why", va[40], "resid", va[41], "lam0", va[42], "trace/frob diff", va[43], va[44], "few phases", [round((st[i]-st[50])/100.0,1) for i in range(50,56)]); print(tag, "stamps(us)", [round(x, 1) for x in t], "sweeps", va[20], "emax", [va[i] for i in range(4)], "delta", [va[8+i] for i in range(3)])
print("loewdin n=%d: %.1f us" % (n, timeit(lambda: ops.loewdin(S, h))))
dbg("loewdin")
St, one, two = make_trdms(4, T, 3)
rng = np.random.default_rng(5)
H = rng.standard_normal((T, T)); H = 0.5 * (H + H.T)
P = T * (T + 1) // 2
a, b = np.tril_indices(T)
rows1 = torch.from_numpy(np.zeros(T * T)).to(dev)
rows2 = torch.from_numpy(np.ascontiguousarray(H[a, b])).to(dev)
Sd = torch.from_numpy(St).to(dev)
print("subspace T=%d: %.1f us" % (T, timeit(lambda: ops.subspace_solve(rows1, rows2, Sd, 2, 1))))
dbg("subspace")
