"""Layer shapes of the other BASELINE.json configs as parity cases on the MI355X (bit-exact against the oracle,
through the C ABI): AlexNet @ batch 32 (fc, m = 32), ResNet-18 @ 256, VGG-16 @ 512 (wide fc rows of W, first
convs with very long rows of U), EfficientNet-B1 @ 1024 in sparse-GPFQ mode (L1, 2-bit; depthwise and
squeeze-excite layers).  Column counts are truncated so that the CPU oracle finishes in seconds; m, groups and
the plan are the real ones."""
import numpy as np
import pytest
import torch

import bench_workload as bw
import golden_inputs as gi

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = [
    # name, N, d(truncated), m, groups, bits, reg, lamb, expected plan prefix
    ("alexnet_fc6_b32", 96, 200, 32, 1, 4, None, 0.1, "resident"),                    # m = B = 32: one partial segment
    ("alexnet_conv2_b32", 48, 150, bw.conv_m(32, 27, 5, 2), 1, 4, None, 0.1, "resident"),
    ("resnet18_l1_conv_b256", 64, 96, bw.conv_m(256, 56, 3, 1), 1, 4, None, 0.1, "coop"),   # m = 23296
    ("resnet18_fc_b256", 1000, 64, 256, 1, 4, None, 0.1, "resident"),
    ("vgg16_conv1_2_b512", 16, 40, bw.conv_m(512, 224, 3, 1), 1, 4, None, 0.1, "coop RT=12 C=128 waves=6 S=704 grid=256 pipel=1"),   # m = 720 384: 704 segments, 5.5 per member: twelve rows x 128 members (round 5; before: the LDS-staged four-row kernel on 64 members -- tests/test_gpu_rounds.py keeps it covered)
    ("vgg16_fc6_b512", 128, 256, 512, 1, 4, None, 0.1, "resident"),
    ("effnet_b1_depthwise_b1024", 8, 9, bw.conv_m(1024, 112, 3, 1), 8, 2, "L1", 0.1, "coop RT=1 C=32 waves=12 S=362 grid=256 rounds=1 groups=8"),   # m = 370 688, N_g = 1
    ("effnet_b1_depthwise5_b1024", 12, 25, bw.conv_m(1024, 14, 5, 2), 12, 2, "L1", 0.1, "resident"),
    ("effnet_b1_se_reduce_b1024", 24, 96, 1024, 1, 2, "L1", 0.1, "resident"),         # 1x1 conv on a 1x1 map: m = B
    ("effnet_b1_project_b1024", 40, 60, bw.conv_m(1024, 28, 1, 0), 1, 2, "L1", 0.1, "coop RT=8 C=32 waves=7 S=197 grid=160 pipe=1"),    # m = 201 728: the pipelined kernel, eight rows x 32 members, one wave for both reducer roles (round 4; before: four rows x 13 sweep waves, columns through LDS -- tests/test_gpu_rounds.py keeps that kernel covered)
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_config_layer_shape_bit_exact(oracle_mod, case):
    from quantized_neural_nets_amd import StepAlgorithm as SA, _lib
    name, N, d, m, groups, bits, reg, lamb, plan = case
    K = 2 ** (bits - 1)
    spec = dict(name=name, N=N, d=d, m=m, bits=bits, scalar=1.16, percentile=1.0, reg=reg, lamb=lamb, groups=groups,
                first_layer=False, zero_every=11, seed=21)
    W, A, X = gi.make_inputs(spec)
    assert _lib.describe_plan(N, d, m, groups).startswith(plan), _lib.describe_plan(N, d, m, groups)
    r = SA._quantize_layer_ex(torch.from_numpy(W).to(DEV), torch.from_numpy(A).to(DEV), torch.from_numpy(X).to(DEV), m,
                              1.16 / K, K, 1.0, reg, lamb, groups, False, torch.device(DEV))
    torch.cuda.synchronize()
    o = oracle_mod.quantize_layer(W, A, X, 1.16 / K, K, 1.0, reg, lamb, groups)
    assert np.float32(float(r["step"])) == o["step"]
    assert np.array_equal(r["idx"].cpu().numpy().astype(np.int16), o["idx"])
    assert np.array_equal(r["U"].cpu().numpy(), o["U"])
    assert np.array_equal(r["Q"].cpu().numpy().view(np.uint32), o["Q"].view(np.uint32))
    assert abs(float(r["quantize_error"]) - o["quantize_error"]) <= 2e-4 * o["quantize_error"]
    assert abs(float(r["relative_quantize_error"]) - o["relative_quantize_error"]) <= 2e-4 * o["relative_quantize_error"]
    if reg == "L1":                                   # sparse GPFQ really produces zeros
        assert float((r["idx"] == 0).float().mean()) > 0.2
    if "rounds=" in _lib.describe_plan(N, d, m, groups):   # the streaming family at the same shape (what these rows ran on before)
        st = SA._quantize_layer_ex(torch.from_numpy(W).to(DEV), torch.from_numpy(A).to(DEV), torch.from_numpy(X).to(DEV), m,
                                   1.16 / K, K, 1.0, reg, lamb, groups, False, torch.device(DEV), plan=_lib.PLAN_STREAM)
        assert torch.equal(st["idx"], r["idx"]) and torch.equal(st["U"], r["U"])
