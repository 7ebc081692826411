"""Repeated launches of every pipelined kernel must reproduce their first result bit for bit (tools/race_screen.py): a
fragment read that beats its LDS-DMA, or a DMA that overwrites a slot still being read, shows up as run-to-run differences."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pipelined_kernels_are_bitwise_reproducible():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "race_screen.py"), "12"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "TOTAL differing runs: 0" in r.stdout
