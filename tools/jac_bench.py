import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, ''+__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))+'')
from evcont_amd import ops, _lib
lib = _lib.load()
dev = torch.device('cuda:0')
for n in (10, 20, 30, 58):
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, n)); S = B @ B.T / n + np.eye(n)
    Sd = torch.from_numpy(S).to(dev)
    for _ in range(3): ops.loewdin(Sd)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.loewdin(Sd)
    e1.record(); torch.cuda.synchronize()
    print(n, "loewdin us:", e0.elapsed_time(e1) / 20 * 1e3)
