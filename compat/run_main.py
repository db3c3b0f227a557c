"""Runs the reference's UNEDITED `main.py` with the MI355X modules in front of its own.

    cd /path/to/Quantized_Neural_Nets/src
    python /path/to/this/repo/compat/run_main.py -model resnet50 -b 4 -bs 1024 -s 1.16

Why a launcher: `python main.py` puts the script's directory (src/) at sys.path[0], AHEAD of PYTHONPATH, so
`from quantize_neural_net import ...` (main.py:8) would still bind the reference's own file.  Started through this
file, sys.path[0] is compat/ instead; the working directory (src/) follows it, so `data_loaders` (main.py:10) and
main.py itself are the reference's, while `quantize_neural_net`, `step_algorithm` and `utils` are the shims here.
No logic beyond that: the arguments go to main.py untouched."""
import os
import runpy
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.getcwd()
sys.path[:] = [_HERE, _SRC] + [p for p in sys.path if os.path.abspath(p or _SRC) not in (_HERE, _SRC)]
_MAIN = os.path.join(_SRC, "main.py")
if not os.path.isfile(_MAIN):
    sys.exit("run_main.py: no main.py in the working directory %s (cd to the reference's src/ first)" % _SRC)
sys.argv[0] = _MAIN
runpy.run_path(_MAIN, run_name="__main__")
