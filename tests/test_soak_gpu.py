"""GPU: determinism soak (tools/soak.py) -- the LDS ring / barrier protocols of the persistent kernels, the key-split
attention and the tiled GEMMs must give bit-identical outputs launch after launch."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_repeated_launches_are_bit_identical():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "2"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    print(r.stdout[-2000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "soak: OK" in r.stdout
