"""Time the cols GEMV (K8 / K7) alone on given shapes: python tools/micro/cols_time.py rows cols [rows cols ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from evcont_amd import ops
dev = torch.device("cuda:0")
args = [int(x) for x in sys.argv[1:]] or [10000, 784, 5050, 82621, 400, 900, 210, 108345]
for rows, cols in zip(args[0::2], args[1::2]):
    A = ops.padded_matrix(torch.randn(rows, cols, device=dev, dtype=torch.float64))
    w = torch.randn(rows, device=dev, dtype=torch.float64)
    out = ops.gemv_cols(A, cols, w)
    err = float((out - w @ A[:, :cols]).abs().max())
    for _ in range(3): ops.gemv_cols(A, cols, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.gemv_cols(A, cols, w)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"rows={rows} cols={cols}: {us:8.1f} us  {rows * cols * 8 / us / 1e6:5.2f} TB/s  max err {err:.1e}", flush=True)
