"""One geometry per step, resident inputs: eager enqueue against the same step replayed as a HIP graph (the Loewdin step is
one kernel under capture).   python tools/micro/graph_exp.py [H30|H10]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from evcont_amd.evaluator import DeviceTRDMs, DeviceAOBatch, BatchedEvaluator
from evcont_amd.synthetic import make_device_ao, make_device_trdm_rows

wl = sys.argv[1] if len(sys.argv) > 1 else "H30"
n, A, T = (30, 30, 20) if wl == "H30" else (10, 10, 5)
dev = torch.device("cuda:0")
S, one, rows = make_device_trdm_rows(n, T, 2, 11, dev)
trd = DeviceTRDMs.from_device_rows(one, rows, S, 2)
trd.compress_sym8_()
del rows
ao = DeviceAOBatch.stack([make_device_ao(n, A, 100, dev, None, ip1_rs_symmetric=True).packed_ip1(eri=True)])
st = torch.cuda.Stream(dev)
ev = BatchedEvaluator(trd, A, 1, stream=st)


def rate(fn, nsteps):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(nsteps):
            fn()
        torch.cuda.synchronize()
        best = max(best, nsteps / (time.perf_counter() - t0))
    return best


print("eager   %9.0f" % rate(lambda: ev.enqueue(ao), 2000), flush=True)
ev.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=st):
    ev.enqueue(ao)
def replay():
    with torch.cuda.stream(st):
        g.replay()
print("graph   %9.0f" % rate(replay, 2000), flush=True)
