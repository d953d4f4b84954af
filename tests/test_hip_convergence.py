"""GPU: the training loop (tools/train_demo.py = the loop body of train.py:245-330) actually learns: a freshly initialised
model fitted to rays rendered from a synthetic scene gains > 15 dB of held-out PSNR in 300 iterations."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_training_converges_on_synthetic_scene():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "train_demo.py"), "--iters", "300", "--n-voxel", str(60 ** 3),
                          "--batch", "2048", "--pool", "65536"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    log = json.loads(out.stdout.strip().splitlines()[-1])["log"]
    assert log[0]["test_psnr"] < 12.0
    assert log[-1]["test_psnr"] > log[0]["test_psnr"] + 15.0, log
    psnrs = [l["test_psnr"] for l in log]
    assert all(b > a - 0.5 for a, b in zip(psnrs, psnrs[1:])), psnrs  # monotone up to noise
