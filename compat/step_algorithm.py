"""Top-level name `step_algorithm`, as the reference's quantize_neural_net.py:9 imports it
(`from step_algorithm import StepAlgorithm`).  No logic: re-exports the MI355X package's class."""
import _locate  # noqa: F401
from quantized_neural_nets_amd.step_algorithm import StepAlgorithm  # noqa: F401
