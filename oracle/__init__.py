"""CPU oracle for the GPFQ hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product package (quantized_neural_nets_amd) never does.
"""
from .gpfq_oracle import (  # noqa: F401
    MODE_MSQ, MODE_SOFT, MODE_HARD, MODE_STOCHASTIC,
    build, quantizer_vec, quantization, quantize_layer, cdot, alphabet_step, max_threads, philox_uniform_vec,
    torch_restatement_quantization,
)
