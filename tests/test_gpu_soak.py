"""Randomized parity soak on the MI355X: tools/soak.py draws random (N, d, m, groups, quantizer, bits, plan) cases over
all four kernel families and compares idx, Q and U bit for bit with the CPU oracle (a longer run of the same tool --
thousands of cases -- is how new kernels are shaken out)."""
import importlib.util
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_random_cases_bit_exact(seed, monkeypatch, capsys):
    spec = importlib.util.spec_from_file_location("soak", os.path.join(ROOT, "tools", "soak.py"))
    soak = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(soak)
    monkeypatch.setattr(sys, "argv", ["soak.py", "60", str(seed)])
    assert soak.main() == 0
    assert "60 cases, 0 mismatches" in capsys.readouterr().out
