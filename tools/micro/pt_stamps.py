"""Phase stamps of one pt_kernel workgroup (build with EVC_DEBUG_STAMPS=1): python tools/micro/pt_stamps.py [energy_only]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np, torch
from evcont_amd import _lib
from evcont_amd.evaluator import DeviceTRDMs, BatchedEvaluator, DeviceAOBatch
from evcont_amd.synthetic import make_device_ao, make_device_trdm_rows
dev = torch.device("cuda:0")
n, A, T, G = 30, 30, 20, 32
eo = len(sys.argv) > 1 and sys.argv[1] == "1"
rows = T * (T + 1) // 2
S_train, one, two_rows = make_device_trdm_rows(n, T, 2, 1, dev, (0, rows))
t = DeviceTRDMs.from_device_rows(one, two_rows, S_train, 2, 0, rows)
del two_rows
t.compress_sym8_()
aos = [make_device_ao(n, A, 10 + k, dev, None, ip1_rs_symmetric=True).packed_ip1(eri=True) for k in range(G)]
aob = DeviceAOBatch.stack(aos)
ev = BatchedEvaluator(t, A, G)
for _ in range(5):
    ev.enqueue(aob, 1, eo)
torch.cuda.synchronize()
lib = _lib.load()
st = (C.c_longlong * 256)()
fn = lib.evc_debug_read_pt; fn.restype = C.c_int; fn.argtypes = [C.c_void_p]
assert fn(st) == 0
names = {0: "entry", 1: "X staged", 2: "xf+frag0"}
for k in range(6):
    b = 3 + 8 * k
    names.update({b: "t%d A mfma" % k, b + 1: "t%d A frag" % k, b + 2: "t%d B mfma" % k, b + 3: "t%d B frag" % k,
                  b + 4: "t%d barrier1" % k, b + 5: "t%d stores issued" % k, b + 6: "t%d barrier2" % k})
for h, nm in ((44, "t1 A"), (49, "t1 B")):
    names.update({h: nm + " start", h + 1: nm + " fetch issued", h + 2: nm + " H issued", h + 3: nm + " N issued", h + 4: nm + " staged"})
if os.environ.get("EVC_PT_PIPE", "1") != "0":
    names = {0: "entry", 1: "X staged", 2: "prologue done"}
    for i in range(12):
        names.update({3 + 3 * i: "i%d start" % i, 4 + 3 * i: "i%d H done" % i, 5 + 3 * i: "i%d barrier" % i})
    names.update({40 + k: "i3 N group %d" % k for k in range(8)})
t0 = min(st[w * 64] for w in range(4))
for i in range(60):
    vals = [st[w * 64 + i] for w in range(4)]
    if all(v == 0 for v in vals): continue
    print("%-18s" % names.get(i, i), "  ".join("%7.2f" % ((v - t0) / 100.0) for v in vals))
wg = (C.c_longlong * (4096 * 3))()
fn = lib.evc_debug_read_pt_wg; fn.restype = C.c_int; fn.argtypes = [C.c_void_p]
assert fn(wg) == 0
w = np.array(wg[:]).reshape(4096, 3)
nwg = 15 * G
w = w[:nwg]
t0 = w[:, 0].min()
st_ = (w[:, 0] - t0) / 100.0; en = (w[:, 1] - t0) / 100.0
print("workgroups", nwg, "first start 0, last start %.2f, first end %.2f, last end %.2f" % (st_.max(), en.min(), en.max()))
print("start offsets percentiles", np.percentile(st_, [0, 10, 50, 90, 99, 100]).round(2))
print("durations percentiles", np.percentile(en - st_, [0, 10, 50, 90, 99, 100]).round(2))
xcc = (w[:, 2] >> 32) & 0xF; hw = w[:, 2] & 0xFFFFFFFF
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
key = xcc * 1000 + se * 100 + sh * 16 + cu
u, cnt = np.unique(key, return_counts=True)
print("distinct CUs used", len(u), "workgroups per CU histogram", np.bincount(cnt))
late = st_ > 5
print("late starters", late.sum(), "their durations", np.percentile((en - st_)[late], [0, 50, 100]).round(2) if late.any() else "")
for L in (0, 1, 8, 100, 240, 256, 300, 479):
    print(L, "xcc", xcc[L], "se", se[L], "cu", cu[L], "start %.2f end %.2f" % (st_[L], en[L]))
dur = en - st_
L = np.arange(nwg); bx = L % 15; by = L // 15
occ = np.array([cnt[np.searchsorted(u, k)] for k in key])
print("mean duration: alone on its CU %.1f (n=%d), sharing %.1f" % (dur[occ == 1].mean(), (occ == 1).sum(), dur[occ == 2].mean()))
print("by xcc", [round(float(dur[(xcc == x) & (occ == 2)].mean()), 1) for x in range(8)])
print("by blockIdx.x", [round(float(dur[(bx == x) & (occ == 2)].mean()), 1) for x in range(15)])
print("by blockIdx.y", [round(float(dur[(by == y) & (occ == 2)].mean()), 1) for y in range(G)])
# partner analysis: the two workgroups of a CU
for k in u[cnt == 2][:12]:
    ii = np.nonzero(key == k)[0]
    print("CU", k, "wgs", ii, "durations", dur[ii].round(1))
