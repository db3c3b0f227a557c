"""Deterministic inputs for the golden GPFQ cases (shared by tools/make_golden.py and the tests).

Inputs are NOT stored in the fixtures: they are regenerated from numpy's PCG64 streams (stable across
numpy versions and platforms) and checked against the sha256 recorded in each fixture.
"""
import hashlib
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# name -> parameters.  bits -> K = 2**(bits-1); step_size (the "step_base" of the driver) = scalar / K.
# kind "loop": one call of StepAlgorithm._quantize_layer (+ per-group _quantization for U).
CASES = {}


def _add(name, N, d, m, bits=4, scalar=1.16, percentile=1.0, reg=None, lamb=0.0, groups=1, first_layer=False,
         zero_every=0, seed=0):
    CASES[name] = dict(name=name, N=N, d=d, m=m, bits=bits, scalar=scalar, percentile=percentile, reg=reg,
                       lamb=lamb, groups=groups, first_layer=first_layer, zero_every=zero_every, seed=seed)


# G2: loop parity, shapes x modes x bits
for (_N, _d, _m) in [(8, 27, 16), (16, 64, 96), (64, 147, 512), (32, 288, 1024)]:
    for _mode, _reg, _lamb in [("msq", None, 0.0), ("soft", "L1", 0.01), ("hard", "L0", 0.01)]:
        for _bits in (2, 4):
            _add("g2_%dx%dx%d_%s_b%d" % (_N, _d, _m, _mode, _bits), _N, _d, _m, bits=_bits, reg=_reg, lamb=_lamb,
                 first_layer=(_d == 27 or _d == 147))
# multi-segment rows (m > 1024, not a multiple of 1024) and many second-level lanes
_add("g2_24x96x2500_msq_b4", 24, 96, 2500)
_add("g2_8x48x5000_soft_b2", 8, 48, 5000, bits=2, reg="L1", lamb=0.05)
_add("g2_4x40x3072_msq_b3", 4, 40, 3072, bits=3)
# G3: dead columns, clipping-heavy alphabet, percentile < 1
_add("g3_zero_cols", 16, 70, 200, zero_every=7)
_add("g3_clip_heavy", 16, 64, 128, scalar=0.3)
_add("g3_percentile95", 16, 80, 160, percentile=0.95)
_add("g3_percentile95_hard", 12, 60, 150, percentile=0.95, reg="L0", lamb=0.02, bits=3)
_add("g3_m1", 6, 20, 1)
_add("g3_N1_d1", 1, 1, 33)
# G4: grouped convolutions (W [N, d_g], A/X [m, groups*d_g])
_add("g4_groups2", 16, 36, 120, groups=2)
_add("g4_groups4_soft", 8, 18, 64, groups=4, reg="L1", lamb=0.01, bits=2)
_add("g4_depthwise", 12, 9, 300, groups=12)
_add("g4_depthwise_hard", 6, 25, 90, groups=6, reg="L0", lamb=0.01)


# G6: headline-scale cases, generated WITHOUT any seed search (seed_offset is always 0): whatever near-ties the
# reference's BLAS order produces are part of the fixture, and tests/tie_audit.py decides what a mismatch means.
# The fixtures store int8 indices + per-row checksums of U instead of Q / U (tools/make_golden.py gen_big_case).
BIG_CASES = {}


def _add_big(name, N, d, m, **kw):
    _add(name, N, d, m, **kw)
    BIG_CASES[name] = CASES.pop(name)


_add_big("g6_256x1152x2048_msq_b4", 256, 1152, 2048)                       # SURVEY.md 7 hard-part 1 probe shape
_add_big("g6_64x576x66560_msq_b4", 64, 576, 66560)                         # 65-segment rows: cooperative plan on auto
_add_big("g6_128x1152x26624_msq_b4", 128, 1152, 26624)                     # ResNet-50 layer2.1-3.conv2 at batch 1024, full size
_add_big("g6_512x1024x3072_msq_b4", 512, 1024, 3072)                       # layer4.1-2.conv2 rows, first 1024 of 4608 columns
_add_big("g6_96x864x4096_soft_b2", 96, 864, 4096, bits=2, reg="L1", lamb=0.1)   # EfficientNet config (2-bit, L1)
_add_big("g6_48x300x5000_hard_b3", 48, 300, 5000, bits=3, reg="L0", lamb=0.02)
# long rows, more of them than the chip holds at once: the cooperative plan in rounds (ResNet-50's 1x1 convs at batch 1024)
_add_big("g6_136x24x51200_msq_b4", 136, 24, 51200)                         # 50 segments, two rounds of 128 rows (16-wave two-row kernel)
_add_big("g6_11x10x803840_msq_b4", 11, 10, 803840)                         # 785 segments, 64 members, 128 granules, two rounds of 8 rows
_add_big("g6_40x12x201728_soft_b2", 40, 12, 201728, bits=2, reg="L1", lamb=0.1)   # 197 segments, 32 members x 4 rows, two rounds
# one row on many workgroups (EfficientNet-B1's first layers at batch 1024) and depthwise rows long enough for the grouped
# cooperative kernel (one row per group, every group its own 9 columns)
_add_big("g6_3x4x3212288_msq_b4", 3, 4, 3212288)                            # 3137 segments: 256 members, four gathered per lane
_add_big("g6_12x6x1440768_soft_b2", 12, 6, 1440768, bits=2, reg="L1", lamb=0.1)    # 1407 segments: 128 members, two per lane
_add_big("g6_dw24x9x100352_soft_b2", 24, 9, 100352, bits=2, reg="L1", lamb=0.1, groups=24)   # depthwise, 98 segments per group


def make_inputs(case, seed_offset=0):
    """W [N, d], A, X [m, groups*d] float32, as SURVEY.md 8(d) prescribes for synthetic activations."""
    c = case
    rng = np.random.default_rng(1234567 + 1000 * c["seed"] + seed_offset + (hash_name(c["name"]) % 100000))
    N, d, m, g = c["N"], c["d"], c["m"], c["groups"]
    D = g * d
    W = (rng.standard_normal((N, d)) * np.sqrt(2.0 / d)).astype(np.float32)
    pre = rng.standard_normal((m, D)).astype(np.float32)
    if c["first_layer"]:
        A = pre.copy()
        X = pre.copy()
    else:
        A = np.maximum(pre, 0).astype(np.float32)
        X = np.maximum(pre + np.float32(0.05) * rng.standard_normal((m, D)).astype(np.float32), 0).astype(np.float32)
    if c["zero_every"]:
        X[:, ::c["zero_every"]] = 0.0
    return W, A, X


def hash_name(name):
    return int(hashlib.sha256(name.encode()).hexdigest()[:8], 16)


def inputs_digest(W, A, X):
    h = hashlib.sha256()
    for a in (W, A, X):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def load_case(name):
    """Returns (case dict, inputs (W, A, X), fixture npz dict).  Verifies the input digest."""
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    fx = dict(np.load(path, allow_pickle=False))
    meta = json.loads(str(fx["meta"]))
    case = meta["case"]
    W, A, X = make_inputs(case, meta.get("seed_offset", 0))
    assert inputs_digest(W, A, X) == meta["inputs_sha256"], "golden input generator drifted for " + name
    return case, (W, A, X), fx, meta


def row_checksums(U):
    """Per-row digests of a residual matrix: crc32 of the fp32 bytes (bit equality), float64 sum and sum of
    squares (tolerance checks), and the first 16 entries."""
    import zlib
    U = np.ascontiguousarray(U, dtype=np.float32)
    crc = np.array([zlib.crc32(U[i].tobytes()) for i in range(U.shape[0])], dtype=np.uint32)
    U64 = U.astype(np.float64)
    return dict(U_crc32=crc, U_sum=U64.sum(1), U_sumsq=(U64 * U64).sum(1), U_head=U[:, :16].copy())


def load_big_case(name):
    """Returns (case dict, inputs (W, A, X), fixture npz dict, meta) for a G6 case.  Verifies the input digest."""
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    fx = dict(np.load(path, allow_pickle=False))
    meta = json.loads(str(fx["meta"]))
    case = meta["case"]
    W, A, X = make_inputs(case, 0)
    assert inputs_digest(W, A, X) == meta["inputs_sha256"], "golden input generator drifted for " + name
    return case, (W, A, X), fx, meta


def available_big_cases():
    return sorted(n for n in BIG_CASES if os.path.exists(os.path.join(GOLDEN_DIR, n + ".npz")))


def available_cases():
    return sorted(n for n in CASES if os.path.exists(os.path.join(GOLDEN_DIR, n + ".npz")))


# ---- G5: the toy network of the driver-level fixture (tools/make_golden.py gen_driver) ---------------------
DRIVER_CONFIGS = [dict(bits=4, reg=None, lamb=0.1, retain_rate=0.25),
                  dict(bits=2, reg='L1', lamb=0.02, retain_rate=0.5),
                  dict(bits=3, reg=None, lamb=0.1, retain_rate=1)]


def toy_net(rng):
    """conv, grouped strided conv, nested Sequential with a 1x1 conv, two Linear layers; weights from rng."""
    import torch
    import torch.nn as nn
    net = nn.Sequential(
        nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(),
        nn.Conv2d(8, 8, 3, stride=2, padding=1, groups=2), nn.ReLU(),
        nn.Sequential(nn.Conv2d(8, 6, 1), nn.ReLU()),
        nn.Flatten(), nn.Linear(6 * 6 * 6, 10), nn.ReLU(), nn.Linear(10, 4))
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(torch.from_numpy((rng.standard_normal(tuple(p.shape)) * 0.3).astype(np.float32)))
    return net.eval()


def toy_batches(rng, B, n):
    import torch
    return [(torch.from_numpy(rng.standard_normal((B, 3, 12, 12)).astype(np.float32)), torch.zeros(B))
            for _ in range(n)]
