"""Time the quarter / four-index transforms alone: python tools/micro/qt_time.py [n ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from evcont_amd import ops
dev = torch.device("cuda:0")
for n in [int(x) for x in sys.argv[1:]] or [40, 48, 58, 64]:
    g = torch.Generator(device=dev).manual_seed(n)
    t = torch.randn(n, n, n, n, generator=g, device=dev, dtype=torch.float64)
    C = torch.randn(n, n, generator=g, device=dev, dtype=torch.float64)
    ref = torch.einsum("abcd,dq->qabc", t, C)
    for tr in (False, True):
        out = ops.quarter_transform(t, C, tr)
        want = ref if not tr else torch.einsum("abcd,qd->qabc", t, C)
        err = float((out - want).abs().max())
        for _ in range(3): ops.quarter_transform(t, C, tr)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.quarter_transform(t, C, tr)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        fl = 2.0 * n ** 5
        print(f"n={n} ct={int(tr)}: {us:8.1f} us  {fl / us / 1e6:6.2f} TFLOP/s  {2 * 8 * n ** 4 / us / 1e6:5.2f} TB/s  max err {err:.1e}", flush=True)
