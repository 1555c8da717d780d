"""pcie_probe.py -- scratch: pageable against pinned copies at the sizes of one 16384-query host call (run on the GPU box)"""
import time

import torch

dev = torch.device("cuda", 0)
for name, nbytes in (("queries 16384x128 f64", 16384 * 128 * 8), ("answers 16384x100 (f64 + i32)", 16384 * 100 * 12), ("half", 8192 * 100 * 12)):
    for pin in (False, True):
        hsrc = torch.empty(nbytes, dtype=torch.uint8, pin_memory=pin)
        hsrc.fill_(1)
        d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        for direction in ("h2d", "d2h"):
            ts = []
            for _ in range(12):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if direction == "h2d":
                    d.copy_(hsrc, non_blocking=True)
                else:
                    hsrc.copy_(d, non_blocking=True)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            ts.sort()
            med = ts[len(ts) // 2]
            print(f"{name:32s} {'pinned  ' if pin else 'pageable'} {direction}: {med * 1e3:7.3f} ms  {nbytes / med / 1e9:6.1f} GB/s")
# host memcpy rate (what staging through a pinned buffer costs)
a = torch.empty(16384 * 128 * 8, dtype=torch.uint8)
a.fill_(2)
b = torch.empty(16384 * 128 * 8, dtype=torch.uint8, pin_memory=True)
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    b.copy_(a)
    ts.append(time.perf_counter() - t0)
ts.sort()
print(f"host memcpy pageable -> pinned 16.8 MB: {ts[5] * 1e3:.3f} ms  {a.numel() / ts[5] / 1e9:.1f} GB/s (torch copy_, {torch.get_num_threads()} threads)")
