"""MI355X-native GPFQ (greedy path-following quantization) hot path.

Drop-in for the per-layer quantization loop of YixuanSeanZhou/Quantized_Neural_Nets
(src/step_algorithm.py + the _quantize_layer driver in src/quantize_neural_net.py): the module names and
call surface of the reference are mirrored here (StepAlgorithm, QuantizeNeuralNet, extract_layers,
InterruptException), the arithmetic runs in hand-written gfx950 HIP kernels behind the C ABI of
include/gpfq.h.  There is no CPU path: everything fails loudly without the built library and a GPU.
"""
from . import _lib  # noqa: F401
from .step_algorithm import StepAlgorithm  # noqa: F401
from .utils import InterruptException, extract_layers  # noqa: F401
from .quantize_neural_net import QuantizeNeuralNet  # noqa: F401

__all__ = ["StepAlgorithm", "QuantizeNeuralNet", "InterruptException", "extract_layers"]
