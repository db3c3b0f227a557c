"""The reference's main.py runs unchanged: `compat/` holds three top-level modules named as the reference names its
own (`quantize_neural_net`, `step_algorithm`, `utils`), so that `main.py:8-9` -- and `data_loaders.py:12`,
`quantize_neural_net.py:9-10` -- resolve without editing a line.  Run in a child interpreter whose sys.path holds
ONLY compat/ (plus the standard library and site-packages): the import lines are executed as text."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMPAT = os.path.join(ROOT, "compat")

# /root/reference/src/main.py:8-9, data_loaders.py:12, quantize_neural_net.py:9-10 -- verbatim import statements
IMPORT_LINES = """\
from quantize_neural_net import QuantizeNeuralNet
from utils import test_accuracy, eval_sparsity, fusion_layers_inplace
from utils import parse_imagenet_val_labels
from step_algorithm import StepAlgorithm
from utils import extract_layers, InterruptException
"""


def _run(code):
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)
    return subprocess.run([sys.executable, "-I", "-c", code], capture_output=True, text=True, env=env, cwd="/tmp", timeout=300)


def test_reference_import_lines_resolve_with_only_compat_on_path():
    code = textwrap.dedent("""
        import sys, inspect
        sys.path.insert(0, %r)
        assert not any(p.rstrip('/') == %r for p in sys.path)       # the repo root is NOT on the path
        exec(%r)
        import quantized_neural_nets_amd as pkg
        assert QuantizeNeuralNet is pkg.QuantizeNeuralNet and StepAlgorithm is pkg.StepAlgorithm
        # the reference calls the operator unbound on the class (quantize_neural_net.py:150, :180)
        assert callable(StepAlgorithm._quantize_layer) and callable(StepAlgorithm._quantization)
        sig = list(inspect.signature(StepAlgorithm._quantize_layer).parameters)
        assert sig == ['W', 'analog_layer_input', 'quantized_layer_input', 'm', 'step_size', 'boundary_idx', 'percentile',
                       'reg', 'lamb', 'groups', 'stochastic_quantization', 'device'], sig
        sig = list(inspect.signature(QuantizeNeuralNet.__init__).parameters)[1:]
        assert sig == ['network_to_quantize', 'network_name', 'batch_size', 'data_loader', 'mlp_bits', 'cnn_bits',
                       'ignore_layers', 'mlp_alphabet_scalar', 'cnn_alphabet_scalar', 'mlp_percentile', 'cnn_percentile',
                       'reg', 'lamb', 'retain_rate', 'stochastic_quantization', 'device'], sig
        for f in (test_accuracy, eval_sparsity, fusion_layers_inplace, parse_imagenet_val_labels, extract_layers):
            assert callable(f)
        assert issubclass(InterruptException, Exception)
        import quantize_neural_net, step_algorithm, utils
        for mod in (quantize_neural_net, step_algorithm, utils):
            assert mod.__file__.startswith(%r), mod.__file__
        print('ok')
    """) % (COMPAT, ROOT, IMPORT_LINES, COMPAT)
    r = _run(code)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_documented_command_binds_compat_ahead_of_same_named_src_modules(tmp_path):
    """INTEGRATION.md A, as documented: `cd <reference>/src; python <repo>/compat/run_main.py ...`.  A stand-in src/ holds
    modules with the reference's names (quantize_neural_net, step_algorithm, utils -- each would raise if imported), a
    data_loaders.py with the reference's own import line (data_loaders.py:12) and a main.py made of the reference's import
    lines (main.py:8-10): the shims must win for the three names, src/ must still provide data_loaders and main.py, and
    the arguments must reach main.py untouched.  (`PYTHONPATH=compat python main.py` does NOT do this: the script's
    directory precedes PYTHONPATH -- checked below as well, so that the documentation cannot drift back.)"""
    src = tmp_path / "src"
    src.mkdir()
    for name in ("quantize_neural_net", "step_algorithm", "utils"):
        (src / (name + ".py")).write_text("raise ImportError('the src/ module %s was imported, not the compat shim')\n" % name)
    (src / "data_loaders.py").write_text("from utils import parse_imagenet_val_labels\n"
                                         "def data_loader(*a, **k):\n    return 'src-data-loader'\n")
    (src / "main.py").write_text(textwrap.dedent("""
        import sys
        from quantize_neural_net import QuantizeNeuralNet
        from utils import test_accuracy, eval_sparsity, fusion_layers_inplace
        from data_loaders import data_loader
        import quantize_neural_net, utils, step_algorithm, data_loaders
        import quantized_neural_nets_amd as pkg
        assert QuantizeNeuralNet is pkg.QuantizeNeuralNet
        assert __name__ == '__main__' and sys.argv[1:] == ['-model', 'alexnet', '-b', '4'], sys.argv
        print('FILES', quantize_neural_net.__file__, utils.__file__, step_algorithm.__file__, data_loaders.__file__, __file__)
    """))
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, os.path.join(COMPAT, "run_main.py"), "-model", "alexnet", "-b", "4"],
                       capture_output=True, text=True, env=env, cwd=str(src), timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    files = r.stdout.strip().splitlines()[-1].split()[1:]
    assert all(f.startswith(COMPAT) for f in files[:3]), files
    assert files[3].startswith(str(src)) and files[4].startswith(str(src)), files
    # the tempting shorter command binds src/'s own modules (here: they raise) -- which is why the launcher exists
    env["PYTHONPATH"] = COMPAT
    r = subprocess.run([sys.executable, "main.py", "-model", "alexnet", "-b", "4"], capture_output=True, text=True, env=env,
                       cwd=str(src), timeout=300)
    assert r.returncode != 0 and "not the compat shim" in r.stderr


def test_compat_modules_hold_no_logic():
    """every shim is imports and a docstring: no def, no class, no control flow"""
    import ast
    for name in ("quantize_neural_net.py", "step_algorithm.py", "utils.py"):
        tree = ast.parse(open(os.path.join(COMPAT, name)).read())
        for node in tree.body:
            assert isinstance(node, (ast.Import, ast.ImportFrom)) or (isinstance(node, ast.Expr) and isinstance(node.value, ast.Constant)), (name, ast.dump(node))
