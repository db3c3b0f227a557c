"""Top-level name `utils`, as the reference's main.py:9, quantize_neural_net.py:10 and data_loaders.py:12 import it
(`from utils import test_accuracy, eval_sparsity, fusion_layers_inplace`, `... extract_layers, InterruptException`,
`... parse_imagenet_val_labels`).  No logic: re-exports the MI355X package's helpers."""
import _locate  # noqa: F401
from quantized_neural_nets_amd.utils import (  # noqa: F401
    SUPPORTED_BLOCK_TYPE, SUPPORTED_LAYER_TYPE, InterruptException, eval_sparsity, extract_layers,
    fusion_layers_inplace, parse_imagenet_val_labels, test_accuracy)
