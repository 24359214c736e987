"""GPU suite: the example in examples/ runs and its built-in assertions hold."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_quickstart_example(capsys):
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "quickstart.py")
    spec = importlib.util.spec_from_file_location("quickstart", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main() == 0
    out = capsys.readouterr().out
    assert "cluster labels: [0, 0, 2, 0, 4]" in out and "substring dedup: B" in out
