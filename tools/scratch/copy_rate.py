#!/usr/bin/env python3
"""What a plain device copy and torch's own transpose reach on this box, next to tools/prep_bench.py's TB/s (moved bytes = read + written)."""
import torch
dev = torch.device("cuda:0")
def rate(fn, nbytes, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return nbytes / (e0.elapsed_time(e1) / n * 1e-3) / 1e12
for (m, D) in [(93184, 576), (93184, 1152), (26624, 2304), (7168, 4608)]:
    x = torch.randn(m, D, device=dev); y = torch.empty_like(x); yt = torch.empty(D, m, device=dev)
    nb = 2 * x.numel() * 4
    print("m=%d D=%d (%.0f MB moved): copy_ %.2f TB/s   transpose (yt.copy_(x.t())) %.2f TB/s" % (
        m, D, nb / 1e6, rate(lambda: y.copy_(x), nb), rate(lambda: yt.copy_(x.t()), nb)))
