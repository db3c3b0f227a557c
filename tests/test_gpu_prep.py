"""Column preparation in one pass (gpfq_prepare_columns_ws_f32: transpose + pad + canonical column norms carried along)
against the two-pass path it replaces (gpfq_prepare_columns_f32: transpose, then gpfq_colnorm_kernel over XT) and against
the CPU oracle's canonical dot product: AT / XT equal, padding zero, nrm2 pairs bit-identical -- for every load mode
(flat, 16-byte tiled, guarded 4-byte), ragged tails, strided and misaligned inputs, one to 3137 segments."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _prep(A, X, m, D, lda, ldx, fused):
    from quantized_neural_nets_amd import _lib
    L = _lib.lib
    mp = L.gpfq_padded_m(m)
    AT = torch.full((max(D, 1), mp), float("nan"), device=DEV)
    XT = torch.full((max(D, 1), mp), float("nan"), device=DEV)
    nrm = torch.full((2 * max(D, 1),), float("nan"), device=DEV)
    st = _lib.current_stream_ptr(torch.device(DEV))
    if fused:
        part = torch.empty((max(int(L.gpfq_prepare_ws_bytes(D, m)), 4) // 4,), device=DEV)
        _lib.check(L.gpfq_prepare_columns_ws_f32(_p(A), lda, _p(X), ldx, m, D, _p(AT), _p(XT), _p(nrm), mp, _p(part), part.numel() * 4, st))
    else:
        _lib.check(L.gpfq_prepare_columns_f32(_p(A), lda, _p(X), ldx, m, D, _p(AT), _p(XT), _p(nrm), mp, st))
    torch.cuda.synchronize()
    return AT, XT, nrm


CASES = [
    # m, D, extra leading dimension, element offset of the base (misalignment), why
    (1, 1, 0, 0, "one sample, one column"),
    (100, 3, 0, 0, "flat, D = 3"),
    (1024, 27, 0, 0, "flat, first conv of a net (27 features), exactly one segment"),
    (1025, 60, 0, 0, "flat, two segments, the second with one sample"),
    (5000, 64, 0, 0, "flat at its widest"),
    (3000, 64, 4, 0, "64 columns but strided rows: tiled, 16-byte loads"),
    (2500, 65, 0, 0, "65 columns: tiled, guarded 4-byte loads (D % 4 != 0), second tile of one column"),
    (7168, 576, 0, 0, "ResNet-50 layer4.0-sized tile count, 16-byte loads"),
    (2048, 147, 0, 0, "147 features (7x7x3 stem): guarded loads, ragged last tile"),
    (4096, 128, 0, 1, "base misaligned by one element: guarded loads"),
    (3333, 40, 0, 2, "flat shape but misaligned base: tiled guarded loads"),
    (66000, 100, 12, 0, "65 segments, strided, ragged"),
    (2048, 1100, 0, 0, "35 tiles of 32 columns in three groups of 12 (one slot of the last group empty), ragged last tile"),
    (3000, 2310, 2, 0, "73 tiles in five groups of 15 (two empty slots), strided, guarded loads"),
    (66000, 2100, 0, 0, "enough workgroups for 64-column tiles: 33 tiles in five groups of 7 (two empty slots)"),
    (1100000, 8, 0, 0, "1075 segments: 32 slots per lane in the slot tree"),
    (3212288, 4, 0, 0, "3137 segments (EfficientNet-B1's 112 x 112 maps at batch 1024)"),
]


@pytest.mark.parametrize("m,D,pad,off,why", CASES, ids=["m%d_D%d_pad%d_off%d" % c[:4] for c in CASES])
def test_fused_prep_equals_two_pass(oracle_mod, m, D, pad, off, why):
    g = torch.Generator().manual_seed(m * 131 + D)
    ld = D + pad
    bufA = torch.randn(m * ld + off + 8, generator=g)
    bufX = torch.relu(torch.randn(m * ld + off + 8, generator=g))
    bufA, bufX = bufA.to(DEV), bufX.to(DEV)
    A = bufA[off:off + m * ld].view(m, ld)
    X = bufX[off:off + m * ld].view(m, ld)
    if D > 2:
        X[:, 1] = 0.0                                     # a zero column: norm 0, reciprocal 0
    AT0, XT0, n0 = _prep(A, X, m, D, ld, ld, fused=False)
    AT1, XT1, n1 = _prep(A, X, m, D, ld, ld, fused=True)
    assert torch.equal(AT1, AT0) and torch.equal(XT1, XT0)
    assert torch.equal(AT1[:, :m], A[:, :D].T) and torch.equal(XT1[:, :m], X[:, :D].T)
    assert not AT1[:, m:].any() and not XT1[:, m:].any()                       # zero padding, no NaN left
    assert torch.equal(n1.view(torch.int32), n0.view(torch.int32)), why       # bit-identical, -0 / NaN included
    # and the oracle's canonical dot product for a few columns: (sqrt(cdot(x, x)))^2
    xs = X[:, :D].cpu().numpy()
    for t in sorted({0, 1 if D > 2 else 0, D // 2, D - 1}):
        r = np.sqrt(oracle_mod.cdot(xs[:, t], xs[:, t]), dtype=np.float32)
        n2 = np.float32(r * r)
        assert np.float32(n1[2 * t].item()) == n2, (t, why)
        assert np.float32(n1[2 * t + 1].item()) == (np.float32(1.0) / n2 if n2 > 0 else np.float32(0.0))


def test_fused_prep_without_workspace_is_the_two_pass_path():
    from quantized_neural_nets_amd import _lib
    L = _lib.lib
    m, D = 3000, 96
    A = torch.randn(m, D, device=DEV)
    X = torch.randn(m, D, device=DEV)
    mp = L.gpfq_padded_m(m)
    AT, XT, nrm = torch.empty(D, mp, device=DEV), torch.empty(D, mp, device=DEV), torch.empty(2 * D, device=DEV)
    _lib.check(L.gpfq_prepare_columns_ws_f32(_p(A), D, _p(X), D, m, D, _p(AT), _p(XT), _p(nrm), mp, None, 0,
                                             _lib.current_stream_ptr(torch.device(DEV))))
    AT0, XT0, n0 = _prep(A, X, m, D, D, D, fused=False)
    assert torch.equal(AT, AT0) and torch.equal(XT, XT0) and torch.equal(nrm, n0)
    assert L.gpfq_prepare_ws_bytes(D, m) == 96 * 3 * 4 + (256 - 96 * 3 * 4 % 256) % 256
