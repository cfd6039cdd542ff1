"""Where the hosted step spends its time (pieces synchronised one by one): python tools/micro/hosted_pieces.py [H2O|Zundel|H10|H30]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from evcont_amd.evaluator import DeviceTRDMs
from evcont_amd.hosted import HostedEvaluator
from evcont_amd.synthetic import make_device_ao, make_device_trdm_rows
W = {"H30": (30, 30, 20, None), "H10": (10, 10, 5, None), "H2O": (13, 3, 10, (9, 2, 2)),
     "Zundel": (28, 7, 30, (9, 2, 2, 2, 9, 2, 2))}
dev = torch.device("cuda:0")
for name in sys.argv[1:] or ["H2O", "H10"]:
    n, A, T, sizes = W[name]
    S, one, rows = make_device_trdm_rows(n, T, 2, 1236, dev)
    trd = DeviceTRDMs.from_device_rows(one, rows, S, 2).compress_sym8_()
    src = make_device_ao(n, A, 5, dev, sizes, ip1_rs_symmetric=True).packed_ip1(eri=True)
    hv = HostedEvaluator(trd, A, src.aoslices.cpu().numpy(), warm_start=False, zero_copy=False)
    st = hv.staging()
    for nm in ("S", "hcore", "ipovlp", "dhcore", "gnuc", "eri", "eri_ip1"):
        np.copyto(st[nm], getattr(src, nm).cpu().numpy().reshape(st[nm].shape))
    for _ in range(3): hv.run()
    main, side = hv.stream, hv.side
    (h0, d0), (h2, d2) = hv._slabs
    acc = {}
    def tick(tag, f):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize()
        acc[tag] = acc.get(tag, 0.0) + (time.perf_counter() - t0)
    def up0():
        with torch.cuda.stream(main): d0.copy_(h0, non_blocking=True)
    def up2():
        side.wait_stream(main)
        with torch.cuda.stream(side): d2.copy_(h2, non_blocking=True)
    def energy():
        with torch.cuda.stream(main): hv.ev.enqueue(hv.aob, 1, energy_only=True)
    def grad():
        with torch.cuda.stream(main):
            main.wait_stream(side); hv.ev.phase_gradient(hv.aob, False)
    def down():
        with torch.cuda.stream(main): hv._out_slab.copy_(hv.ev.energy_grad, non_blocking=True)
    for _ in range(20):
        tick("upload early", up0); tick("upload late (forked)", up2); tick("energy phase", energy); tick("gradient phase", grad); tick("download", down)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): hv.run()
    whole = (time.perf_counter() - t0) / 20
    print(name, {k: round(v / 20 * 1e6) for k, v in acc.items()}, "whole step", round(whole * 1e6), "us;",
          "slab bytes", h0.numel() * 8, h2.numel() * 8, flush=True)
