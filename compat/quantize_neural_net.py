"""Top-level name `quantize_neural_net`, as the reference's main.py:8 imports it
(`from quantize_neural_net import QuantizeNeuralNet`).  No logic: re-exports the MI355X package's module."""
import _locate  # noqa: F401
from quantized_neural_nets_amd.quantize_neural_net import (  # noqa: F401
    CONV2D_MODULE_TYPE, LAYER_LOGGING, LINEAR_MODULE_TYPE, RESULT_LOGGING_DIR, QuantizeNeuralNet, SaveInputConv2d,
    SaveInputMLP)
from quantized_neural_nets_amd.step_algorithm import StepAlgorithm  # noqa: F401  (quantize_neural_net.py:9)
from quantized_neural_nets_amd.utils import InterruptException, extract_layers  # noqa: F401  (quantize_neural_net.py:10)
