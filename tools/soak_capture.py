#!/usr/bin/env python3
"""Randomized soak of the conv capture (gpfq_gather_patches_f32 straight into the column layout) against the unfold path
of quantize_neural_net.py:334-347 in torch ops (GPU box only; test infrastructure, not product).  Random
(B, C, H, W, kernel, padding, dilation, retain rate, memory format) cases must agree bit for bit, padding included.
    python tools/soak_capture.py [cases=300] [seed=1]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quantized_neural_nets_amd.quantize_neural_net as qnn  # noqa: E402
from quantized_neural_nets_amd.step_algorithm import PreparedColumns  # noqa: E402
from quantized_neural_nets_amd.utils import InterruptException  # noqa: E402


def capture(fused, x, k, pad, dil, retain, seed):
    qnn.FUSED_CAPTURE = fused
    np.random.seed(seed)
    hook = qnn.SaveInputConv2d(kernel_size=k, dilation=dil, padding=pad, stride=1, groups=1, retain_rate=retain)
    for xin in (x, x * 2.0):
        try:
            hook(None, (xin,), None)
        except InterruptException:
            pass
    return hook.inputs


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    t0 = time.time()
    bad = 0
    kinds = {}
    for ci in range(ncases):
        kh = int(rng.choice([1, 2, 3, 3, 3, 4, 5, 5, 6, 7, 7]))
        kw = kh if rng.integers(0, 4) else int(rng.choice([1, 2, 3, 4, 5, 6, 7]))
        dil = (int(rng.choice([1, 1, 1, 2])), int(rng.choice([1, 1, 1, 2, 3])))
        pad = (int(rng.integers(0, kh + 1)), int(rng.integers(0, kw + 1)))
        B, C = int(rng.integers(1, 7)), int(rng.integers(1, 90))
        H = int(rng.integers(dil[0] * (kh - 1) + 1, dil[0] * (kh - 1) + 40))
        W = int(rng.integers(dil[1] * (kw - 1) + 1, dil[1] * (kw - 1) + 40))
        retain = float(rng.choice([0.1, 0.25, 0.5, 0.9, 1.0]))
        g = torch.Generator().manual_seed(int(rng.integers(0, 1 << 30)))
        x = torch.randn(B, C, H, W, generator=g).to(dev)
        fmt = int(rng.integers(0, 3))
        if fmt == 1:
            x = x.contiguous(memory_format=torch.channels_last)
        elif fmt == 2:                              # a view with a storage offset and a padded row stride
            big = torch.randn(B, C, H + 1, W + 3, generator=g).to(dev)
            x = big[:, :, 1:, 2:W + 2]
        s = int(rng.integers(0, 1 << 30))
        a = capture(True, x, (kh, kw), pad, dil, retain, s)
        b = capture(False, x, (kh, kw), pad, dil, retain, s)
        qnn.FUSED_CAPTURE = True
        ok = True
        for u, v in zip(a, b):
            ok = ok and isinstance(u, PreparedColumns) and tuple(u.shape) == tuple(v.shape) and torch.equal(u.matrix(), v) \
                and u.T.shape[1] % 1024 == 0 and float(u.T[:, u.m:].abs().sum()) == 0.0
        key = "%dx%d" % (kh, kw) if kh == kw and dil == (1, 1) else "other"
        kinds[key] = kinds.get(key, 0) + 1
        if not ok:
            bad += 1
            print("MISMATCH", dict(B=B, C=C, H=H, W=W, k=(kh, kw), pad=pad, dil=dil, retain=retain, fmt=fmt, seed=s), flush=True)
        if (ci + 1) % 50 == 0:
            print("%d cases, %d mismatches, %.0f s" % (ci + 1, bad, time.time() - t0), flush=True)
    print("capture soak: %d cases (%s), %d mismatches" % (ncases, ", ".join("%s: %d" % kv for kv in sorted(kinds.items())), bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
