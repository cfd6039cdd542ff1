"""Rates of the four ways a caller drives the H30 pipeline (one process = one setting of the environment):
MD regime (batch 1), one stream (batch 32), three caller streams, one caller stream pipelined inside the library.
    python tools/micro/split_exp.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from evcont_amd.evaluator import DeviceTRDMs, DeviceAOBatch, BatchedEvaluator, PipelinedBatchedEvaluator
from evcont_amd.synthetic import make_device_ao, make_device_trdm_rows

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
n, A, T, G = 30, 30, 20, 32
S, one, rows = make_device_trdm_rows(n, T, 2, 11, dev)
trd = DeviceTRDMs.from_device_rows(one, rows, S, 2)
trd.compress_sym8_()
del rows
aos = [make_device_ao(n, A, 100 + k, dev, None, ip1_rs_symmetric=True).packed_ip1(eri=True) for k in range(64)]
b32 = [DeviceAOBatch.stack(aos[0:32]), DeviceAOBatch.stack(aos[32:64])]
b1 = [DeviceAOBatch.stack([a]) for a in aos[:8]]


def rate(fn, nsteps, per):
    for k in range(10):
        fn(k)
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        for k in range(nsteps):
            fn(k)
        torch.cuda.synchronize()
        best = max(best, nsteps * per / (time.perf_counter() - t0))
    return best


skip_md = bool(os.environ.get("SKIP_MD"))   # (an MD evaluator leaves the device's side stream behind: a fifth stream)
if not skip_md:
    e1 = BatchedEvaluator(trd, A, 1)
    print("md_regime      %9.0f" % rate(lambda k: e1.enqueue(b1[k % 8]), steps * 4, 1), flush=True)
e32 = BatchedEvaluator(trd, A, G)
print("single_stream  %9.0f" % rate(lambda k: e32.enqueue(b32[k % 2]), steps, G), flush=True)
sts = [torch.cuda.Stream(dev) for _ in range(3)]
es = [BatchedEvaluator(trd, A, G, stream=s) for s in sts]
print("three_streams  %9.0f" % rate(lambda k: es[k % 3].enqueue(b32[k % 2]), steps, G), flush=True)
pe = PipelinedBatchedEvaluator(trd, A, G, depth=3)
tk = []
def pstep(k):
    if len(tk) == 3:
        pe.results(tk.pop(0))
    tk.append(pe.enqueue(b32[k % 2]))
print("pipelined      %9.0f" % rate(pstep, steps, G), flush=True)
if not skip_md:
    ew = BatchedEvaluator(trd, A, 1, warm_start=True)
    print("md_warm        %9.0f" % rate(lambda k: ew.enqueue(b1[0]), steps * 4, 1), flush=True)
