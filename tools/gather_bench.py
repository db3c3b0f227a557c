#!/usr/bin/env python3
"""Time gpfq_gather_patches_f32 (the fused conv activation capture, quantize_neural_net.py:334-347 of the reference) on
ResNet-50 layer shapes at batch 1024 (GPU box only).   python tools/gather_bench.py"""
import ctypes, sys, time
import numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantized_neural_nets_amd import _lib
dev = torch.device('cuda:0')
for (B, C, H, k, pad) in [(1024, 64, 56, 3, 1), (1024, 128, 28, 3, 1), (1024, 256, 14, 3, 1), (1024, 512, 7, 3, 1), (256, 3, 224, 7, 3)]:
    x = torch.randn(B, C, H, H, device=dev)
    L1 = (H + 2 * pad - (k - 1) - 1) // k + 1
    L = L1 * L1
    keep = int(0.25 * L + 1)
    sel = torch.from_numpy(np.concatenate([L * i + np.random.choice(L, size=keep) for i in range(B)])).to(dev)
    m = sel.numel(); mp = _lib.lib.gpfq_padded_m(m)
    T = torch.empty((C * k * k, mp), device=dev)
    st = _lib.current_stream_ptr(dev)
    def run():
        _lib.check(_lib.lib.gpfq_gather_patches_f32(ctypes.c_void_p(x.data_ptr()), B, C, H, H, k, k, pad, pad, 1, 1,
                   ctypes.c_void_p(sel.data_ptr()), m, ctypes.c_void_p(T.data_ptr()), mp, st))
    run(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    out_bytes = T.numel() * 4
    print("B=%d C=%d H=%d k=%d: m=%d d=%d  %.3f ms  out %.1f MB -> %.0f GB/s written" % (B, C, H, k, m, C*k*k, ms, out_bytes/1e6, out_bytes/ms/1e6))
    if os.environ.get("GATHER_SORTED"):
        # upper bound of what reading the patches of a 64-column block in position order could give (round 4 experiment: the
        # columns come out in sorted order here, which the product could only undo with a lane permutation per value)
        blk = torch.arange(m, device=dev) // 64
        order = torch.argsort(blk * (1 << 40) + sel, stable=True)
        sel_sorted = sel[order].contiguous()
        sel_keep, sel = sel, sel_sorted
        run(); torch.cuda.synchronize()
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        print("    patches sorted by position inside every 64-column block: %.3f ms" % (e0.elapsed_time(e1) / 10))
        sel = sel_keep
