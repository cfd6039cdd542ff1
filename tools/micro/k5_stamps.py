"""Per-workgroup timeline of the batched K5 kernel (build with EVC_DEBUG_STAMPS=1): python tools/micro/k5_stamps.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np, torch
from evcont_amd import _lib
from evcont_amd.evaluator import DeviceTRDMs, BatchedEvaluator, DeviceAOBatch
from evcont_amd.synthetic import make_device_ao, make_device_trdm_rows
dev = torch.device("cuda:0")
n, A, T, G = 30, 30, 20, 32
rows = T * (T + 1) // 2
S_train, one, two_rows = make_device_trdm_rows(n, T, 2, 1, dev, (0, rows))
t = DeviceTRDMs.from_device_rows(one, two_rows, S_train, 2, 0, rows)
del two_rows
t.compress_sym8_()
aos = [make_device_ao(n, A, 10 + k, dev, None, ip1_rs_symmetric=True).packed_ip1(eri=True) for k in range(G)]
aob = DeviceAOBatch.stack(aos)
ev = BatchedEvaluator(t, A, G)
for _ in range(5):
    ev.enqueue(aob, 1, True)
torch.cuda.synchronize()
lib = _lib.load()
LDS = os.environ.get("EVC_ROWS_LDS") != "0"
wg = (C.c_longlong * (8192 if LDS else 4096))()
fn = lib.evc_debug_read_k5 if os.environ.get("EVC_ROWS_LDS") == "0" else lib.evc_debug_read_k5l; fn.restype = C.c_int; fn.argtypes = [C.c_void_p]
assert fn(wg) == 0
w = np.array(wg[:]).reshape(1024, 8 if LDS else 4)
w = w[w[:, 0] != 0]
live = w[:, 1] != 0      # blocks that returned early have no later stamps
w = w[live]
t0 = w[:, 0].min()
st, lp, en = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0, (w[:, 2] - t0) / 100.0
print("workgroups", len(w), "start pct", np.percentile(st, [0, 50, 90, 100]).round(2))
print("main loop (us) pct", np.percentile(lp - st, [0, 10, 50, 90, 100]).round(2))
print("epilogue (us) pct", np.percentile(en - lp, [0, 50, 100]).round(2), "last end", en.max().round(2))
xcc = (w[:, 3] >> 32) & 0xF
print("by xcc: main loop mean", [round(float((lp - st)[xcc == x].mean()), 1) for x in range(8)], "count", [int((xcc == x).sum()) for x in range(8)])
hw = w[:, 3] & 0xFFFFFFFF
key = xcc * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xF)
u, cnt = np.unique(key, return_counts=True)
print("distinct CUs", len(u), "wgs per CU histogram", np.bincount(cnt))

if LDS:
    ep = (w[:, 4:8] - w[:, 1:2]) / 100.0
    print("epilogue stamps after the wave-0 loop end (us, median): barrier", np.median(ep[:, 0]).round(2), "LDS writes", np.median(ep[:, 1]).round(2),
          "barrier", np.median(ep[:, 2]).round(2), "pass 0 done", np.median(ep[:, 3]).round(2), "; p90", np.percentile(ep, 90, axis=0).round(2))
