"""Makes `quantized_neural_nets_amd` importable when ONLY this directory is on sys.path (the reference's
main.py is run from its own src/ directory through `python <repo>/compat/run_main.py`): the package lives one level up."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.append(_ROOT)
