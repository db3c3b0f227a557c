"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/gpfq.h declares, the
argument validation that needs no device, plan selection, the neuron-shard partition and the index->value
rebuild used after the all_gather."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    so = os.path.join(ROOT, "quantized_neural_nets_amd", "libgpfq_hip.so")
    if not os.path.exists(so):
        ge.build()
    from quantized_neural_nets_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "gpfq.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(gpfq_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 10
    raw = ctypes.CDLL(lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), "libgpfq_hip.so does not export " + name
    assert declared == set(lib.EXPORTS), (declared ^ set(lib.EXPORTS))
    assert lib.lib.gpfq_abi_version() == 3


def test_padding_and_workspace(lib):
    assert lib.lib.gpfq_padded_m(1) == 1024 and lib.lib.gpfq_padded_m(1024) == 1024
    assert lib.lib.gpfq_padded_m(1025) == 2048 and lib.lib.gpfq_padded_m(93184) == 93184
    ws = lib.lib.gpfq_workspace_bytes(64, 576, 93184, 1)
    assert ws >= 2 * 576 * 93184 * 4 + 576 * 4


def test_host_side_argument_errors_need_no_gpu(lib):
    L = lib.lib
    one = ctypes.c_void_p(256)     # never dereferenced: validation fails first
    assert L.gpfq_quantizer_f32(7, 0.1, one, 4, 8, 0.0, None, one, None, None) == -1
    assert L.gpfq_quantizer_f32(0, 0.1, None, 4, 8, 0.0, None, one, None, None) == -1
    assert L.gpfq_row_absmax_f32(one, 3, 4, 8, one, None) == -1                      # ldw < d
    assert L.gpfq_prepare_columns_f32(one, 8, one, 8, 100, 8, one, one, one, 512, None) == -1   # m_pad wrong
    assert b"m_pad" in L.gpfq_last_error()
    rc = L.gpfq_quantize_layer_f32(one, one, 32, one, 32, 6, 8, 100, 4, 0.1, 8, 0, 0.0, 0, 0, one, None, 1, one, None, one, 1 << 30, 0, None)
    assert rc == -1 and b"divisible" in L.gpfq_last_error()                           # N % groups
    rc = L.gpfq_quantize_layer_f32(one, one, 8, one, 8, 8, 8, 100, 1, 0.1, 8, 0, 0.0, 0, 0, one, None, 1, one, None, one, 16, 0, None)
    assert rc == -2                                                                   # workspace too small
    rc = L.gpfq_quantize_groups_prepared_f32(one, one, one, one, one, one, 8, 8, 100, 1024, 1, 0.1, 200, 0, 0.0, 0, 0,
                                             one, 1, None, 0, None, 0, None)
    assert rc == -1 and b"int8" in L.gpfq_last_error()                                # K too big for int8


def test_spin_limit_word_can_always_be_exceeded_before_the_counter_wraps(lib, monkeypatch):
    """The kernels bound an exchange with `(spins += 256) > word + pause` on a 32-bit counter (gpfq_loop_kernels.h
    reducer_section, gpfq_pipe_kernels.h pipe_reducer; pause < 32).  At the clamp the counter must still be able to exceed
    the word: with 2^24 - 1 polls it reached 0xFFFFFF00, which is not greater, and wrapped to 0 -- an unbounded spin."""
    L = lib.lib
    for asked, polls in (("16384", 16384), ("0", 0), ("-5", 0), (str((1 << 24) - 2), (1 << 24) - 2),
                         (str((1 << 24) - 1), (1 << 24) - 2), (str(1 << 30), (1 << 24) - 2)):
        monkeypatch.setenv("GPFQ_COOP_SPIN_LIMIT", asked)
        word = L.gpfq_spin_limit_word()
        assert word == 256 * polls, (asked, word)
        for pause in (0, 31):
            limit = word | pause
            # the last value the counter takes before it would wrap: a multiple of 256 below 2^32
            assert 0xFFFFFF00 > limit, (asked, hex(limit))
            # ... and the kernel's loop, replayed from the last few polls before the bound: it gives up without wrapping
            spins, gave_up = max(0, (limit & ~255) - 512), False
            for _ in range(8):
                spins = (spins + 256) & 0xFFFFFFFF
                if spins > limit:
                    gave_up = True
                    break
                assert spins != 0, "the counter wrapped"
            assert gave_up
    monkeypatch.delenv("GPFQ_COOP_SPIN_LIMIT")
    assert L.gpfq_spin_limit_word() == 256 * 16384


def test_cooperative_launch_api_is_off_until_asked_for_or_contention_is_seen(lib, monkeypatch):
    L = lib.lib
    monkeypatch.delenv("GPFQ_COOP_LAUNCH_API", raising=False)
    L.gpfq_clear_contention()
    assert L.gpfq_coop_launch_api_active() == 0          # (the sticky switch itself needs a timeout: tests/test_gpu_pipe.py)
    monkeypatch.setenv("GPFQ_COOP_LAUNCH_API", "1")
    assert L.gpfq_coop_launch_api_active() == 1
    monkeypatch.setenv("GPFQ_COOP_LAUNCH_API", "-1")
    assert L.gpfq_coop_launch_api_active() == 0


def test_plan_selection(lib):
    assert lib.describe_plan(512, 4608, 3072).startswith("resident")
    assert lib.describe_plan(256, 2304, 7168).startswith("resident")
    assert "S=7" in lib.describe_plan(256, 2304, 7168)
    assert lib.describe_plan(64, 576, 93184, 1, lib.PLAN_STREAM).startswith("stream")
    # the pipelined cooperative kernels (round 4) where they are modelled -- and were measured -- faster than the lock-step ones
    assert lib.describe_plan(64, 576, 93184).startswith("coop RT=4 C=16 waves=6 S=91 grid=256 pipe=1")
    assert lib.describe_plan(128, 1152, 93184).startswith("coop RT=8 C=16 waves=6 S=91 grid=256 pipe=1")
    assert lib.describe_plan(128, 1152, 26624).startswith("coop RT=4 C=8 waves=4 S=26 grid=256 pipe=1")  # four single rows x 8 members (1.23-1.36 us per column; lock-step two rows x 4 members 1.46)
    assert lib.describe_plan(64, 576, 23296).startswith("coop RT=2 C=8 waves=3 S=23 grid=256 d=")        # the pipelined choice would have two sweep waves per member: lock-step stays
    assert lib.describe_plan(256, 2304, 26624).startswith("coop RT=8 C=8 waves=4 S=26 grid=256 pipe=1")
    assert lib.describe_plan(8, 576, 93184).startswith("coop RT=1 C=16 waves=6")        # an 8-GPU shard of 64 rows
    assert lib.describe_plan(32, 2304, 26624).startswith("coop RT=1 C=8 waves=4")       # an 8-GPU shard of 256 rows
    assert lib.describe_plan(512, 4608, 13312).startswith("coop RT=8 C=4 waves=4 S=13 grid=256 pipe=1")   # long rows, more rows than CUs
    assert lib.describe_plan(64, 9, 30000, 64).startswith("coop RT=1 C=4 waves=8 S=30 grid=256 rounds=1 groups=64")   # depthwise, long rows: one row per group
    assert lib.describe_plan(96, 9, 370688, 96).startswith("coop RT=1 C=32 waves=12 S=362 grid=256 rounds=12 groups=96")
    assert lib.describe_plan(64, 9, 30000, 32).startswith("stream")                     # two rows per group: not the depthwise case
    # long rows, more of them than the chip holds: cooperative in rounds (one co-resident launch per block of rows)
    # (four rows x 13 sweep waves: the variant that stages its columns through LDS)
    # (eight rows per tile, seven sweep waves + ONE reducer wave for both roles: the pipelined `_w8s` variant)
    assert lib.describe_plan(2048, 1024, 51200).startswith("coop RT=8 C=8 waves=7 S=50 grid=256 rounds=8 pipe=1")
    assert lib.describe_plan(512, 128, 201728).startswith("coop RT=8 C=32 waves=7 S=197 grid=256 rounds=8 pipe=1")
    assert lib.describe_plan(1024, 512, 201728).startswith("coop RT=12 C=32 waves=7 S=197 grid=256 rounds=11 pipel=1")   # 11 rounds of twelve rows against 16 of eight: 35.1 against 37.5-40.5 us per column
    # (round 5: twelve rows per tile in three groups, columns through LDS, 128 members: 11 rounds instead of the 16 of the four-row
    # lock-step kernel on 64 members, which stays selectable -- GPFQ_COOP_PIPEL=0)
    assert lib.describe_plan(256, 64, 803840).startswith("coop RT=12 C=128 waves=7 S=785 grid=256 rounds=11 pipel=1")
    assert lib.describe_plan(512, 256, 803840).startswith("coop RT=12 C=128 waves=7 S=785 grid=256 rounds=22 pipel=1")
    assert lib.describe_plan(64, 147, 263168).startswith("coop RT=12 C=64 waves=5 S=257 grid=256 rounds=2 pipel=1")
    assert lib.describe_plan(256, 512, 13312).startswith("resident RT=1 waves=13")      # <= 16 segments, one round: whole rows, no exchange
    assert lib.describe_plan(16, 32, 3212288).startswith("coop RT=4 C=256 waves=13 S=3137 grid=256 rounds=4")    # four rows on the whole chip (1024 granules)
    assert lib.describe_plan(2, 32, 3212288).startswith("coop RT=2 C=256 waves=13 S=3137 grid=256 d=32")           # two rows: the 512-granule kernel
    assert lib.describe_plan(2048, 512, 13312).startswith("coop RT=8 C=2 waves=7 S=13 grid=256 rounds=2 pipe=1")   # cooperative rounds against resident rounds
    assert lib.describe_plan(1, 32, 3212288).startswith("coop RT=1 C=256 waves=13 S=3137 grid=256 d=32")         # one row on the whole chip
    assert lib.describe_plan(4, 8, 4194304).startswith("stream")                        # 4096 segments: beyond 256 members x 15
    assert lib.describe_plan(1000, 2048, 1024).startswith("resident")
    # the fallback plan: whole rows per workgroup, never an exchange, whatever the shape
    for shape in ((64, 576, 93184), (8, 576, 93184), (512, 4608, 3072), (256, 2304, 803840)):
        desc = lib.describe_plan(*shape, 1, lib.PLAN_STREAM_ROWS)
        assert desc.startswith("stream") and " C=" not in desc, desc
    with pytest.raises(lib.GpfqError):
        lib.describe_plan(8, 8, 4_200_000)


def test_plan_selection_lock_step_family(lib, monkeypatch):
    """GPFQ_COOP_PIPE=0: the cooperative choices of rounds 1-3 (the lock-step kernels stay in the library and stay the choice
    for one or two rows per tile, one row per group, 13 sweep waves and 256+ granules)."""
    monkeypatch.setenv("GPFQ_COOP_PIPE", "0")
    assert lib.describe_plan(64, 576, 93184).startswith("coop RT=2 C=8 waves=12")
    assert lib.describe_plan(128, 1152, 93184).startswith("coop RT=4 C=8 waves=12")
    assert lib.describe_plan(256, 2304, 26624).startswith("coop RT=4 C=4 waves=7")
    assert lib.describe_plan(512, 4608, 13312).startswith("coop RT=4 C=2 waves=7")
    assert lib.describe_plan(2048, 1024, 51200).startswith("coop RT=4 C=4 waves=13 S=50 grid=256 rounds=8")
    assert lib.describe_plan(1024, 512, 201728).startswith("coop RT=4 C=16 waves=13 S=197 grid=256 rounds=16")
    assert lib.describe_plan(2048, 512, 13312).startswith("coop RT=4 C=2 waves=7 S=13 grid=256 rounds=4")
    assert lib.describe_plan(128, 1152, 26624).startswith("coop RT=2 C=4 waves=7 S=26 grid=256 d=")
    assert lib.describe_plan(256, 64, 803840).startswith("coop RT=4 C=64 waves=13 S=785 grid=256 rounds=16")    # 256 granules
    monkeypatch.setenv("GPFQ_COOP_PIPE", "1")        # forced: also where AUTO would not take it
    monkeypatch.setenv("GPFQ_COOP_PIPEL", "0")
    assert lib.describe_plan(64, 576, 23296).startswith("coop RT=4 C=16 waves=2 S=23 grid=256 pipe=1")
    monkeypatch.setenv("GPFQ_COOP_RT", "8")
    assert lib.describe_plan(128, 1152, 26624).startswith("coop RT=8 C=16 waves=2 S=26 grid=256 pipe=1")
    monkeypatch.delenv("GPFQ_COOP_RT")
    monkeypatch.setenv("GPFQ_COOP_C", "128")         # 256 granules per gather: only when asked for
    assert lib.describe_plan(21, 6, 803840).startswith("coop RT=8 C=128 waves=7 S=785 grid=256 rounds=2 pipe=1")
    monkeypatch.delenv("GPFQ_COOP_C")
    monkeypatch.delenv("GPFQ_COOP_PIPE")
    monkeypatch.setenv("GPFQ_COOP_PIPEL", "1")       # the twelve-row family wherever a configuration exists
    assert lib.describe_plan(1024, 256, 51200).startswith("coop RT=12 C=8 waves=7 S=50 grid=256 rounds=3 pipel=1")
    assert lib.describe_plan(512, 128, 201728).startswith("coop RT=12 C=32 waves=7 S=197 grid=256 rounds=6 pipel=1")
    monkeypatch.setenv("GPFQ_COOP_PIPEL", "0")
    assert lib.describe_plan(256, 64, 803840).startswith("coop RT=4 C=64 waves=13 S=785 grid=256 rounds=16")


def test_partition_covers_every_neuron_once():
    from quantized_neural_nets_amd import dist as qd
    for N, groups, world in [(64, 1, 8), (1000, 1, 8), (7, 1, 8), (32, 32, 8), (96, 96, 5), (24, 2, 8), (24, 3, 4),
                             (512, 1, 1), (6, 6, 8)]:
        kind, chunk = qd.partition(N, groups, world)
        seen = np.zeros(N, int)
        Ng = N // groups
        for r in range(world):
            a, b = qd.local_range(kind, chunk, N, groups, r)
            assert 0 <= a <= b
            if kind == "rows":
                seen[a:b] += 1
            elif kind == "groups":
                seen[a * Ng:b * Ng] += 1
            else:
                for g in range(groups):
                    seen[g * Ng + a:g * Ng + b] += 1
        assert (seen == 1).all(), (N, groups, world, kind)
        assert kind == ("rows" if groups == 1 else "groups" if groups >= world else "rows_in_groups")


def test_rebuild_q_matches_the_quantizers(oracle_mod):
    """Q rebuilt from gathered indices == the value the quantizer produced (up to the sign of zero)."""
    from quantized_neural_nets_amd import dist as qd
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(4000) * 0.4).astype(np.float32)
    for mode in (0, 1, 2):
        q, idx = oracle_mod.quantizer_vec(mode, 0.0731, x, 8, 0.05)
        r = qd.rebuild_q(torch.from_numpy(idx.astype(np.int8)), float(np.float32(0.0731)), 8, mode, float(np.float32(0.05)))
        assert np.array_equal(r.numpy(), q)
    un = rng.random(4000).astype(np.float32)
    q, idx = oracle_mod.quantizer_vec(3, 0.0731, x, 8, uniform=un)
    r = qd.rebuild_q(torch.from_numpy(idx.astype(np.int8)), float(np.float32(0.0731)), 8, 3, 0.0)
    assert np.array_equal(r.numpy(), q)


def test_extract_layers_order_and_whitelist():
    import torch.nn as nn
    from quantized_neural_nets_amd.utils import extract_layers, register_block_type

    class Block(nn.Module):
        def __init__(self):
            super().__init__()
            self.c = nn.Conv2d(2, 2, 1)

    net = nn.Sequential(nn.Conv2d(3, 2, 3), nn.ReLU(), nn.Sequential(nn.Linear(4, 4), nn.Conv2d(2, 2, 1)), Block(),
                        nn.Linear(4, 2))
    layers = []
    extract_layers(net, layers)
    assert [type(l).__name__ for l in layers] == ["Conv2d", "Linear", "Conv2d", "Linear"]   # Block is not whitelisted
    register_block_type(Block)
    layers = []
    extract_layers(net, layers)
    assert len(layers) == 5 and layers[3] is net[3].c


def test_bench_workload_matches_survey_totals():
    import bench_workload as bw
    L = [l[:4] for l in bw.resnet50_3x3_layers(1024)]
    assert len(L) == 16 and sum(n * d for _, n, d, _ in L) == 11317248
    assert abs(sum(bw.algorithmic_bytes(n, d, m) for _, n, d, m in L) / 1e12 - 0.837) < 1e-3
    assert {m for *_, m in L} == {93184, 26624, 7168, 3072}
    # the other configs' models: ResNet-18 at batch 256, VGG-16 at batch 512
    R = bw.resnet18_layers(256)
    assert len(R) == 21 and abs(sum(n * d for _, n, d, _ in R) / 1e6 - 11.68) < 0.01
    assert abs(sum(bw.algorithmic_bytes(n, d, m) for _, n, d, m in R) / 1e12 - 0.213) < 2e-3
    assert min(m for *_, m in R) == 256 and max(m for *_, m in R) == 200960
    V = bw.vgg16_layers(512)
    assert len(V) == 16 and abs(sum(n * d for _, n, d, _ in V) / 1e6 - 138.3) < 0.1
    assert abs(sum(bw.algorithmic_bytes(n, d, m) for _, n, d, m in V) / 1e12 - 2.449) < 5e-3
    assert max(m for *_, m in V) == 720384
    # BASELINE.json configs[3] and [4]: all 54 ResNet-50 layers; EfficientNet-B1's 116 layers
    A = bw.normalize_layers(bw.resnet50_all_layers(1024))
    assert len(A) == 54 and abs(sum(l[1] * l[2] for l in A) / 1e6 - 25.50) < 0.01
    assert abs(sum(bw.algorithmic_bytes(l[1], l[2], l[3], l[4]) for l in A) / 1e12 - 7.215) < 5e-3
    E = bw.efficientnet_b1_layers(1024)
    assert len(E) == 116 and abs(sum(l[1] * l[2] for l in E) / 1e6 - 7.72) < 0.01
    assert abs(sum(bw.algorithmic_bytes(l[1], l[2], l[3], l[4]) for l in E) / 1e12 - 1.133) < 1e-3
    dw = [l for l in E if l[4] > 1]
    assert len(dw) == 23 and all(l[1] == l[4] and l[2] in (9, 25) for l in dw)
    assert min(l[3] for l in dw) == 2048 and max(l[3] for l in dw) == 370688
    assert sum(1 for l in E if ".fc" in l[0] and l[3] == 1024) == 46


@pytest.mark.parametrize("K,mode", [(1, 0), (2, 1), (4, 0), (8, 0), (8, 2), (2, 2), (64, 3), (200, 0)])
def test_packed_index_fields_round_trip(K, mode):
    """packed.pack_indices / unpack_indices: n-bit fields for every alphabet size, ragged lengths, range check."""
    from quantized_neural_nets_amd import packed
    lo, count = packed.index_range(K, mode)
    assert count == (2 * (K + 1) + 1 if mode == 2 else 2 * K + 1)
    g = torch.Generator().manual_seed(K * 10 + mode)
    for n in (0, 1, 7, 8, 9, 1000, 4099):
        idx = torch.randint(lo, lo + count, (n,), generator=g)
        if n >= 2:
            idx[0], idx[-1] = lo, lo + count - 1
        data, nbits = packed.pack_indices(idx.to(torch.int16), K, mode)
        assert data.dtype == torch.uint8
        if nbits != 16:
            assert nbits == packed.field_bits(K, mode) and data.numel() == ((n + 7) // 8) * nbits
        back = packed.unpack_indices(data, nbits, n, K, mode)
        assert torch.equal(back, idx)
    with pytest.raises(ValueError):
        packed.pack_indices(torch.tensor([lo + count]), K, mode)


def test_packed_rebuild_equals_oracle_q(oracle_mod):
    """indices -> packed -> indices -> rebuild_q gives the oracle's Q bit for bit (msq, soft and hard alphabets)."""
    import golden_inputs as gi
    from quantized_neural_nets_amd import dist, packed
    for name in ("g2_8x27x16_msq_b4", "g2_8x27x16_soft_b2", "g2_8x27x16_hard_b4", "g2_16x64x96_msq_b2"):
        case = gi.CASES[name]
        W, A, X = gi.make_inputs(case)
        K = 2 ** (case["bits"] - 1)
        o = oracle_mod.quantize_layer(W, A, X, case["scalar"] / K, K, case["percentile"], case["reg"], case["lamb"],
                                      case["groups"])
        mode = 1 if case["reg"] == "L1" else 2 if case["reg"] == "L0" else 0
        idx = torch.from_numpy(o["idx"].astype(np.int64))
        data, nbits = packed.pack_indices(idx, K, mode)
        back = packed.unpack_indices(data, nbits, idx.numel(), K, mode).reshape(idx.shape)
        q = dist.rebuild_q(back, float(o["step"]), K, mode, float(case["lamb"] or 0.0))
        assert np.array_equal(q.numpy(), o["Q"])


def test_torch_extension_registers_the_operators_and_refuses_cpu_tensors(lib):
    """The TORCH_LIBRARY(gpfq) extension over the C ABI (level 3): schema as the survey gives it (plus
    the plan and the fused sum-of-squares output); only the HIP dispatch key exists, so CPU tensors are an error."""
    from quantized_neural_nets_amd import torch_ext
    assert os.path.exists(torch_ext.LIB_PATH)
    schema = str(torch.ops.gpfq.quantize_layer.default._schema)
    assert schema.startswith("gpfq::quantize_layer(Tensor W, Tensor A, Tensor X, float step, int K, int mode, float lamb, int groups, int seed, int plan)")
    assert schema.endswith("-> (Tensor, Tensor, Tensor, Tensor)")
    with pytest.raises((RuntimeError, NotImplementedError)):
        torch.ops.gpfq.quantize_layer(torch.zeros(4, 4), torch.zeros(8, 4), torch.zeros(8, 4), 0.1, 8, 0, 0.0, 1, 0, 0)
    with pytest.raises((RuntimeError, NotImplementedError)):
        torch.ops.gpfq.quantizer(torch.zeros(4), 0.1, 8, 0, 0.0, None)


def test_bench_quotes_only_a_pmc_summary_of_its_own_kernel_sources(tmp_path, monkeypatch):
    """roofline.traffic comes from a committed rocprofv3 PMC summary ONLY if that summary carries the digest of the kernel
    sources the run was built from (tools/pmc_traffic.py stamps it); a summary of other sources yields null, not stale bytes."""
    import json
    import bench
    from quantized_neural_nets_amd import _lib
    digest = _lib.kernel_source_digest()
    assert len(digest) == 64 and digest == _lib.kernel_source_digest()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    kern = {"void gpfq::gpfq_resident_rt2_m0_w8(gpfq::SlabParams)": {"launches": 6, "hbm_bytes_per_launch": 123}}
    (prof / "r99_a_pmc_traffic.json").write_text(json.dumps({"source_sha256": "0" * 64, "kernels": kern}))
    got, src = bench.pmc_traffic("gpfq_resident_rt2_m0_w8", digest)
    assert got is None and "no PMC summary for this kernel source" in src
    (prof / "r99_b_pmc_traffic.json").write_text(json.dumps({"source_sha256": digest, "kernels": kern}))
    got, src = bench.pmc_traffic("gpfq_resident_rt2_m0_w8", digest)
    assert got == 123 and src.endswith("r99_b_pmc_traffic.json")
    # and the committed summary of this round matches the committed sources, kernel names included
    import glob
    import re
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")),       # (r02_v10 after r02_v9)
                    key=lambda f: [int(x) if x.isdigit() else x for x in re.split(r"(\d+)", os.path.basename(f))])[-1]
    real = json.load(open(newest))
    assert real["source_sha256"] == digest, "%s was collected on other kernel sources: re-collect (tools/profile_bench.sh)" % newest
    # (the headline's kernels since round 4: the resident ones and the pipelined cooperative ones -- layer2.{1,2,3}.conv2 moved
    # from the lock-step two-row kernel to four single rows x 8 members late in the round)
    assert any("gpfq_resident_rt2_m0_w8" in k for k in real["kernels"]) and any("gpfq_pipe_rg2_m0_w8" in k for k in real["kernels"])
    assert any("gpfq_pipe_rg1_m0_w8" in k for k in real["kernels"]) and any("gpfq_resident_rt1_m0_w8" in k for k in real["kernels"])
    # ... and so does the counter summary behind roofline_issue / the measured roofline_l2
    newest_c = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_counters.json")),
                      key=lambda f: [int(x) if x.isdigit() else x for x in re.split(r"(\d+)", os.path.basename(f))])[-1]
    cnt = json.load(open(newest_c))
    assert cnt["source_sha256"] == digest, "%s was collected on other kernel sources: re-collect (tools/profile_counters.sh)" % newest_c
    dom = [v for k, v in cnt["kernels"].items() if "gpfq_resident_rt2_m0_w8" in k][0]["per_launch"]
    assert 0.0 < dom["SQ_ACTIVE_INST_ANY"] / dom["SQ_WAVE_CYCLES"] <= 1.0 and dom["TCP_TCC_READ_REQ_sum"] > 0
    # the kernel names bench.py derives from a plan description are the names rocprofv3 reports
    assert bench.kernel_name("resident RT=2 waves=7 S=7 grid=(256,1) d=4608") == "gpfq_resident_rt2_m0_w8"
    assert bench.kernel_name("coop RT=4 C=8 waves=12 S=91 grid=256 d=1152") == "gpfq_coop_rt4_m0_w12"
    assert bench.kernel_name("coop RT=2 C=4 waves=7 S=26 grid=256 d=1152", 1) == "gpfq_coop_rt2_m1_w8"
    assert bench.kernel_name("coop RT=2 C=64 waves=13 S=785 grid=256 rounds=32 d=64") == "gpfq_coop_rt2_m0_w16"
    assert bench.kernel_name("coop RT=4 C=32 waves=9 S=257 grid=256 rounds=2 d=147") == "gpfq_coop_rt4_m0_w12"
    assert bench.kernel_name("coop RT=1 C=256 waves=13 S=3137 grid=256 rounds=16 d=32", 1) == "gpfq_coop_rt1_m1_w16"
    assert bench.kernel_name("coop RT=2 C=256 waves=13 S=3137 grid=256 rounds=8 d=32", 1) == "gpfq_coop_rt2_m1_w16o"
    assert bench.kernel_name("coop RT=4 C=4 waves=13 S=50 grid=256 rounds=8 d=1024") == "gpfq_coop_rt4_m0_w16l"
    assert bench.kernel_name("coop RT=4 C=64 waves=13 S=785 grid=256 rounds=16 d=64", 1) == "gpfq_coop_rt4_m1_w16lq"
    assert bench.kernel_name("coop RT=1 C=16 waves=6 S=91 grid=128 d=576") == "gpfq_coop_rt1_m0_w12"
    assert bench.kernel_name("coop RT=1 C=32 waves=12 S=362 grid=256 rounds=12 groups=96 d=9", 1) == "gpfq_coop_rt1g_m1_w12"
    assert bench.kernel_name("coop RT=12 C=128 waves=7 S=785 grid=256 rounds=11 pipel=1 d=64", 1) == "gpfq_pipel_m1_w8"
    assert bench.plan_rounds("coop RT=2 C=64 waves=13 S=785 grid=256 rounds=32 d=64") == 32
    assert bench.plan_rounds("coop RT=4 C=8 waves=12 S=91 grid=256 d=1152") == 1
    assert bench.l2_column_bytes("resident RT=2 waves=7 S=7 grid=(256,1) d=4608", 512, 4608, 7168) == 256 * 4608 * 8 * 7168
    assert bench.l2_column_bytes("stream RT=4 waves=8 S=50 grid=(320,1) d=320", 1280, 320, 51200) is None
    # one-segment rows of m <= 256 / 512 samples: the kernels load one / two quarters of the padded segment
    assert bench.l2_column_bytes("resident RT=2 waves=1 S=1 grid=(2048,1) d=4096", 4096, 4096, 1024, 1, 512) == 2048 * 4096 * 8 * 512
    assert bench.l2_column_bytes("resident RT=1 waves=1 S=1 grid=(1000,1) d=512", 1000, 512, 1024, 1, 256) == 1000 * 512 * 8 * 256
    assert bench.l2_column_bytes("resident RT=1 waves=1 S=1 grid=(1000,1) d=2048", 1000, 2048, 1024, 1, 1024) == 1000 * 2048 * 8 * 1024


def test_division_free_msq_form_equals_the_division_form(tmp_path):
    """The loop kernels find the MSQ index from ONE multiplication of the dot product (gpfq_device.h quant_msq_from_dot)
    and run the reference's two divisions only when the product lies within (K + 4) * 2^-18 of a rounding boundary.  The
    C restatement of both forms (tests/csrc/msq_fast_check.c, every operation individually rounded) compares them on
    random and boundary-hugging arguments: wherever the fast form claims an answer it must be the division form's, bit
    for bit.  (4.8e9 samples were run once; this pass is 1.6e7.)  The check itself is checked: with the tolerance at
    (K + 4) * 2^-24 it must find mismatches."""
    import subprocess
    src = os.path.join(ROOT, "tests", "csrc", "msq_fast_check.c")
    exe = str(tmp_path / "msq_fast_check")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-o", exe, src, "-lm"], check=True)
    out = subprocess.run([exe, "2000000", "11"], check=True, capture_output=True, text=True).stdout
    assert " 0 mismatches" in out, out
    frac = float(out.split(" samples, ")[1].split()[0])
    assert 0.2 < frac < 0.9, out                     # both sides of the guard are exercised
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-DTOL_UNIT=0x1p-24f", "-o", exe + "_loose", src, "-lm"], check=True)
    loose = subprocess.run([exe + "_loose", "2000000", "11"], capture_output=True, text=True)
    assert loose.returncode == 1 and " 0 mismatches" not in loose.stdout, loose.stdout
    # the device helper uses the constants the check was run with
    dev = open(os.path.join(ROOT, "quantized_neural_nets_amd", "csrc", "gpfq_device.h")).read()
    host = open(os.path.join(ROOT, "quantized_neural_nets_amd", "csrc", "gpfq_capi.hip")).read()
    assert "(p.qc.Kf + 4.0f) * 0x1p-18f" in host and "0x1p-60f" in dev and "p.qc.Kf <= 1024.0f" in host


def test_bench_oracle_shape_check_samples_rows_and_groups(oracle_mod):
    """bench.py's oracle_shape_check on the CPU, with the oracle's own full-layer indices standing in for the GPU's: the
    sampled rows / groups / columns must line up (0 mismatches), one check per DISTINCT shape, and a corrupted index in a
    sampled position (last row of the last group, first column) must be found."""
    import numpy as np
    import torch
    import bench
    import bench_workload as bw
    layers = [("a", 40, 12, 300, 1), ("a2", 40, 12, 300, 1), ("dw", 6, 5, 2000, 6), ("g2", 8, 7, 1500, 2)]
    data, idx = [], {}
    for li, (name, N, dg, m, groups) in enumerate(layers):
        W, A, X = bw.synthetic_layer(N, groups * dg, m, 50 + li, rows_d=dg)
        step = bw.layer_step(W, 1.16, 2)
        o = oracle_mod.quantize_layer(W.numpy(), A.numpy(), X.numpy(), 1.16 / 2, 2, 1.0, "L1", 0.05, groups, step=np.float32(step))
        data.append((name, W, A, X, step, m))
        idx[name] = torch.from_numpy(o["idx"].astype(np.int8))
    rec = bench.oracle_shape_check(data, layers, idx, 2, 1, 0.05, 2e9)
    assert rec["shapes"] == 3 and rec["mismatches"] == 0 and rec["weights"] == 32 * 12 + 4 * 5 + 8 * 7     # (dw: the first two and the last two of its six groups)
    idx["dw"][5, 0] += 1
    rec = bench.oracle_shape_check(data, layers, idx, 2, 1, 0.05, 2e9)
    assert rec["mismatches"] == 1 and rec["failed"] and rec["failed"][0].startswith("dw group 5")
    # the budget bounds rows x columns x m per shape (never below 4 columns)
    rec = bench.oracle_shape_check(data[:1], layers[:1], {"a": idx["a"]}, 2, 1, 0.05, 32 * 300 * 5)
    assert rec["weights"] == 32 * 5 and rec["mismatches"] == 0


def test_bench_counter_rooflines_are_bounded_and_digest_gated(tmp_path, monkeypatch):
    """roofline_issue / the measured roofline_l2 come from a committed counter summary (tools/pmc_counters.py) ONLY when it
    carries the digest of the kernel sources in use; the issue fraction is SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES (<= 1 by
    construction), the L2 figure TCP_TCC_READ_REQ x 128 B over the launch time measured live."""
    import json
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    per = {"SQ_WAVES": 768.0, "SQ_WAVE_CYCLES": 1000.0e6, "SQ_ACTIVE_INST_ANY": 620.0e6, "SQ_ACTIVE_INST_VALU": 460.0e6,
           "SQ_WAIT_ANY": 270.0e6, "SQ_WAIT_INST_ANY": 110.0e6, "SQ_INSTS_VALU": 461.0e6, "SQ_INSTS_SALU": 95.0e6,
           "TCP_TCC_READ_REQ_sum": 226.5e6, "TCC_HIT_sum": 219.7e6, "TCC_MISS_sum": 7.3e6, "TCC_EA0_RDREQ_sum": 7.25e6}
    kern = {"gpfq::gpfq_resident_rt2_m0_w8(gpfq::SlabParams)": {"launches": 6, "per_launch": per, "shapes": {}}}
    (prof / "r99_pmc_counters.json").write_text(json.dumps({"source_sha256": "0" * 64, "kernels": kern}))
    fam = {"ms": 2.244 * 3, "launches": 3}
    got, src = bench.counter_rooflines("gpfq_resident_rt2_m0_w8", fam, "f" * 64, {"frac": 0.45})
    assert got is None and "no counter summary for this kernel source" in src
    got, src = bench.counter_rooflines("gpfq_resident_rt2_m0_w8", fam, "0" * 64, {"frac": 0.45})
    assert src.endswith("r99_pmc_counters.json")
    iss, l2 = got["issue"], got["l2"]
    assert iss["frac"] == 0.62 and iss["valu_frac"] == 0.46 and iss["wait_frac"] == 0.27 and iss["issue_stall_frac"] == 0.11
    assert 0.0 < iss["frac"] <= 1.0 and abs(iss["frac"] + iss["wait_frac"] + iss["issue_stall_frac"] - 1.0) < 1e-9
    assert iss["insts_per_wave"]["valu"] == round(461.0e6 / 768, 1)
    assert l2["measured"] and l2["request_bytes"] == 128 and l2["model"] == {"frac": 0.45}
    assert abs(l2["achieved"] - 226.5e6 * 128 / 2.244e-3 / 1e9) < 0.1 and abs(l2["frac"] - l2["achieved"] / 34500.0) < 1e-4
    assert abs(l2["l2_hit_rate"] - 219.7 / 227.0) < 1e-4 and l2["frac"] <= 1.0


def test_bench_roofline_bound_holds_every_family_against_the_roof_that_binds_it():
    """roofline_bound (round 5): per kernel family a FLOOR on its time -- vector ALU (packed instructions x 4.45 SIMD cycles with two
    waves per SIMD, 5.2 per instruction of a single wave, profiles/r05_probe_valu.txt), vector L1 (64 B per clock per CU), the
    exposed exchange of the lock-step kernels (0.65 us, profiles/r03_xchg_probe.txt), HBM -- summed and divided by the measured
    times: one fraction <= 1 for the whole step."""
    import bench
    us = lambda insts, per_simd: insts * max(5.2, 4.45 * per_simd) / 2.4e9 * 1e6           # noqa: E731
    # twelve rows in three groups, seven sweep waves (two per SIMD): 3 x 160 packed instructions per column and round
    name, ms = bench.roof_floor_ms("coop RT=12 C=128 waves=7 S=785 grid=256 rounds=11 pipel=1 d=64", 64, 1, 1e12, 1e9)
    assert name == "vector ALU" and abs(ms - 64 * 11 * 3 * us(160, 2) * 1e-3) < 1e-9
    # four groups of a row pair, six sweep waves: 4 x 80; of single rows with four sweep waves (one per SIMD: a wave's own issue rate)
    assert abs(bench.roof_floor_ms("coop RT=8 C=16 waves=6 S=91 grid=256 pipe=1 d=1152", 1152, 1, 1e12, 1e9)[1] - 1152 * 4 * us(80, 2) * 1e-3) < 1e-9
    assert abs(bench.roof_floor_ms("coop RT=4 C=8 waves=4 S=26 grid=256 pipe=1 d=1152", 1152, 1, 1e12, 1e9)[1] - 1152 * 4 * us(48, 1) * 1e-3) < 1e-9
    # lock-step: the sweep of all rows, then an exposed exchange, per column and round
    name, ms = bench.roof_floor_ms("coop RT=4 C=64 waves=13 S=785 grid=256 rounds=16 d=64", 64, 1, 1e12, 1e9)
    assert name == "vector ALU + exchange" and abs(ms - 64 * 16 * (us(160, 4) + 0.65) * 1e-3) < 1e-9
    # resident: the larger of the vector-L1 floor and the sweeps' issue floor
    l2b = 256 * 2304 * 8 * 7168
    name, ms = bench.roof_floor_ms("resident RT=1 waves=7 S=7 grid=(256,1) d=2304", 2304, 1, 1e12, l2b)
    assert name == "vector L1" and abs(ms - l2b / 34.5e12 * 1e3) < 1e-9
    name, ms = bench.roof_floor_ms("resident RT=2 waves=3 S=3 grid=(64,1) d=4608", 4608, 1, 1e12, 64 * 4608 * 8 * 3072)
    assert name == "vector ALU" and abs(ms - 4608 * us(80, 1) * 1e-3) < 1e-9       # (a quarter of the CUs busy: the chain, not the L1)
    assert bench.roof_floor_ms("stream RT=4 waves=8 S=50 grid=(64,1) d=10", 10, 1, 8e9, None) == ("hbm", 1.0)
    # ... and the whole-step fraction: time-weighted, <= 1 whenever every family's floor is below its time
    fam = {"a": {"ms": 20.0, "floor_ms": 10.0, "roofs": {"vector ALU": 10.0}}, "b": {"ms": 10.0, "floor_ms": 8.0, "roofs": {"vector L1": 8.0}}}
    rb = bench.roofline_bound(fam, 2.0, 8e9, 10, False)
    assert rb["families"]["column preparation"]["frac"] == 0.5 and rb["families"]["a"]["roof"] == "vector ALU"
    assert abs(rb["frac"] - (1.0 + 0.8 + 1.0) / (2.0 + 1.0 + 2.0)) < 1e-4 and rb["frac"] <= 1.0 and rb["floors_exceeded"] == []
    for src in rb["sources"].values():
        if src.startswith("profiles/"):
            assert os.path.exists(os.path.join(ROOT, src.split(" ")[0])), src


def test_bench_self_launch_builds_the_drivers_command_and_never_touches_the_gpu(monkeypatch, capsys):
    """`python bench.py --gpus N` (N > 1) outside torchrun starts itself under torch.distributed.run as a CHILD process --
    the exact form the driver uses (one rank per GPU, rendezvous on 127.0.0.1), its own flags passed on untouched -- relays
    the child's JSON line and exit code, and does so before anything initialises the GPU (an exec or a HIP call in the
    parent would break the multi-process run on this pool)."""
    import bench
    import torch
    seen = {}

    class FakeProc:
        def __init__(self, cmd, **kw):
            seen["cmd"], seen["kw"] = cmd, kw
            seen["cuda_initialised_at_launch"] = torch.cuda.is_initialized()
            self.stdout = iter(["rank 0 chatter\n", '{"metric": "m", "value": 1.0, "n_gpus": 4}\n'])

        def wait(self):
            return 0

    monkeypatch.setattr(bench.subprocess, "Popen", FakeProc)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2", "--workload", "r50_all"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 0
    cmd = seen["cmd"]
    port = cmd[cmd.index("--master-port") + 1]
    assert cmd == [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr",
                   "127.0.0.1", "--master-port", port, os.path.abspath(bench.__file__),
                   "--gpus", "4", "--steps", "7", "--warmup", "2", "--workload", "r50_all"]
    assert 1024 <= int(port) <= 65535
    env = seen["kw"]["env"]
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert seen["kw"].get("stdout") is not None and not seen["kw"].get("shell")
    assert seen["cuda_initialised_at_launch"] is False and not torch.cuda.is_initialized()
    out = capsys.readouterr()
    assert out.out.strip() == '{"metric": "m", "value": 1.0, "n_gpus": 4}' and "rank 0 chatter" in out.err
    assert bench.launch_command(2, 29500, ["--gpus", "2"])[3:10] == ["--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                                                                      "127.0.0.1", "--master-port", "29500"]


def test_bench_expected_block_reads_the_committed_emulation_lines():
    """`bench.py --gpus N` carries an `expected` block: the per-rank step ONE GPU measured for rank 0's rows of an N-GPU
    world (bench.py --emulate-world N, committed under profiles/), so that SCALE can be held against DESIGN.md 8."""
    import bench
    for wl in ("r50_3x3", "r50_all"):
        for n in (2, 4, 8):
            e = bench.expected_per_rank(wl, n)
            assert e is not None and e["source"].startswith("profiles/") and e["per_rank_ms_per_step"] > 0 and e["value"] > 0
    assert bench.expected_per_rank("r50_3x3", 3) is None
