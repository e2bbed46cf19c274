"""GPU: a slice of tools/soak.py -- random problems of every kernel family and data flow against the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [1, 8])
def test_random_differential_soak(seed):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "soak.py"), "400", str(seed)], capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "0 failures" in out.stdout


def test_random_api_sequences():
    """tools/soak_api.py: random sequences of host / device-pointer / batched evaluations, operator re-uploads that switch
    the data flow, and accessor calls on one context, each checked against the oracle."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_api.py"), "250", "3"], capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " 0 failures" in out.stdout
