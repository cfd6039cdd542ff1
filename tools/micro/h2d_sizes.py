"""Latency of ONE pinned H2D copy (+ synchronize) by size, on the current stream and on a forked one: python tools/micro/h2d_sizes.py"""
import time, torch
dev = torch.device("cuda:0")
side = torch.cuda.Stream(dev)
for nbytes in (15_000, 156_000, 381_280, 1_320_000, 1_800_000, 7_771_008, 10_692_000, 12_500_000):
    n = nbytes // 8
    h = torch.zeros(n, dtype=torch.float64).pin_memory(); d = torch.zeros(n, dtype=torch.float64, device=dev)
    res = []
    for forked in (False, True):
        for _ in range(3):
            d.copy_(h, non_blocking=True); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            if forked:
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    d.copy_(h, non_blocking=True)
                torch.cuda.current_stream(dev).wait_stream(side)
            else:
                d.copy_(h, non_blocking=True)
            torch.cuda.current_stream(dev).synchronize()
        res.append((time.perf_counter() - t0) / 30)
    print(f"{nbytes/1e6:7.3f} MB: same stream {res[0]*1e6:7.1f} us ({nbytes/res[0]/1e9:5.1f} GB/s)   forked {res[1]*1e6:7.1f} us", flush=True)
