"""A fixed-seed slice of the randomised parity sweeps (scripts/fuzz_*.py: random shapes / strides / batch sizes against ATen-CPU) as
part of the GPU suite.  The scripts print every mismatch and end with "<n> problems"."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("script,iters", [("fuzz_parity.py", 240), ("fuzz_conv_bf16.py", 40), ("fuzz_gemm_bf16.py", 30),
                                          ("fuzz_attention.py", 20), ("fuzz_rowops.py", 100), ("fuzz_bn_handoff.py", 12), ("fuzz_bn_backward.py", 10)])
def test_randomised_parity_sweep(gpu, script, iters):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script), "7", str(iters)], capture_output=True, text=True, timeout=600)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " 0 problems" in r.stdout, tail
