import torch, time
n = 3_000_000_000 // 8
x = torch.empty(n, dtype=torch.float64, device="cuda")
y = torch.empty(n, dtype=torch.float64, device="cuda")
def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
ms = t(lambda: x.fill_(1.5)); print(f"fill  (write only) {n*8/ms/1e6:.0f} GB/s  {ms:.3f} ms")
ms = t(lambda: x.zero_()); print(f"zero  (write only) {n*8/ms/1e6:.0f} GB/s  {ms:.3f} ms")
ms = t(lambda: y.copy_(x)); print(f"copy  (r+w)       {2*n*8/ms/1e6:.0f} GB/s  {ms:.3f} ms")
ms = t(lambda: x.sum()); print(f"sum   (read only)  {n*8/ms/1e6:.0f} GB/s  {ms:.3f} ms")
ms = t(lambda: torch.add(x, 1.0, out=y)); print(f"add   (r+w)        {2*n*8/ms/1e6:.0f} GB/s  {ms:.3f} ms")
