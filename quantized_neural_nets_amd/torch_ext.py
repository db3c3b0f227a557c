"""torch.ops.gpfq -- the thin PyTorch-ROCm C++ extension over the C ABI (csrc/gpfq_torch_ext.cpp, SURVEY.md 8(b) level 3).

    from quantized_neural_nets_amd import torch_ext          # registers the operators
    Q, idx, U, usq_seg = torch.ops.gpfq.quantize_layer(W, A, X, step, K, mode, lamb, groups, seed, plan)
    q = torch.ops.gpfq.quantizer(x, step, K, mode, lamb, None)

W (N, d_g), A / X (m, groups*d_g): float32 tensors on the MI355X; `step` is the alphabet step of
step_algorithm.py:191-192 as a host value; mode 0 msq / 1 soft (L1) / 2 hard (L0) / 3 stochastic; plan 0 = auto.
Outputs are allocated by the extension on the inputs' device and the work is queued on torch's current HIP stream.
Only the HIP ("CUDA") dispatch key is registered: CPU tensors raise (there is no CPU path).
The library is built in-tree by csrc/Makefile (`make -C quantized_neural_nets_amd/csrc all`, or
__graft_entry__.build()); importing this module without it raises ImportError.
"""
import os

import torch

from . import _lib  # noqa: F401  (loads libgpfq_hip.so, which the extension links against)

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgpfq_torch.so")

if not os.path.exists(LIB_PATH):
    raise ImportError("quantized_neural_nets_amd.torch_ext: %s is missing; build it with `make -C %s all`" % (
        LIB_PATH, os.path.join(os.path.dirname(LIB_PATH), "csrc")))
torch.ops.load_library(LIB_PATH)

quantize_layer = torch.ops.gpfq.quantize_layer
quantizer = torch.ops.gpfq.quantizer
