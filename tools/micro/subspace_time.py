"""Time the subspace kernels alone: python tools/micro/subspace_time.py [T ...]  (cold, count problems side by side)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from evcont_amd import _lib
from evcont_amd.active_learning import subspace_energies

dev = torch.device("cuda:0")
Ts = [int(x) for x in sys.argv[1:]] or [20, 32, 33, 48, 64, 80, 100, 128, 160]
for T in Ts:
    for count in (1, 32):
        rng = np.random.default_rng(T)
        A = rng.standard_normal((T, T)); S = A @ A.T / T + np.eye(T)
        Hs = []
        for g in range(count):
            B = rng.standard_normal((T, T)); Hs.append(0.5 * (B + B.T) - 3 * np.eye(T))
        H = torch.from_numpy(np.stack(Hs)).to(dev); Sd = torch.from_numpy(S).to(dev)
        for _ in range(3):
            e = subspace_energies(H, Sd)
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        reps = 10
        t0.record()
        for _ in range(reps):
            e = subspace_energies(H, Sd)
        t1.record(); torch.cuda.synchronize()
        import scipy.linalg as sla
        w = sla.eigh(Hs[0], S, eigvals_only=True)
        if os.environ.get("EVC_DEBUG_STAMPS"):
            import ctypes as C
            lib = _lib.load(); st = (C.c_longlong * 16)(); va = (C.c_double * 16)()
            fn = lib.evc_debug_read_big; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
            fn(st, va, 16)
            names = ["chol+inv", "assemble H", "W, C", "sym+shift", "few roots", "jacobi", "tail"]
            print("   ", ", ".join(f"{nm} {(st[i + 1] - st[i]) / 100.0:.1f}" for i, nm in enumerate(names)), f"us; sweeps {va[0]:.0f} few_ok {va[1]:.0f}")
            print("    few: tridiag %.1f bisect %.1f vectors %.1f check %.1f us; worst/scale %.2e lam0 %.12f (scipy %.12f) bracket %.1e" % (
                (st[8] - st[4]) / 100.0, (st[9] - st[8]) / 100.0, (st[10] - st[9]) / 100.0, (st[11] - st[10]) / 100.0, va[2], va[3],
                __import__("scipy.linalg").linalg.eigh(Hs[0], S, eigvals_only=True)[0], va[4]))
        print(f"T={T:4d} count={count:3d}: {t0.elapsed_time(t1) / reps * 1e3:9.1f} us per launch   |dE0|={abs(float(e[0,0]) - w[0]):.2e}", flush=True)
