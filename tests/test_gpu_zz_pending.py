"""Runs the -m gpu test files that were written without GPU access (end of round 2) in a subprocess each, so that a crash in
never-run code is contained; their first results are this test's output.  Non-strict xfail until that first run has been looked
at (tools/first_gpu_check.sh): after it, remove the guards in those files and delete this one."""
import os
import subprocess
import sys

import pytest
import torch

PENDING = ["test_gpu_zz_prompt_generator.py", "test_gpu_zz_prompt_based_segmentation.py", "test_gpu_zz_training_encoders.py"]
HERE = os.path.dirname(os.path.abspath(__file__))

pytestmark = [pytest.mark.gpu,
              pytest.mark.xfail(strict=False, reason="first GPU run pending (written without GPU access at the end of round 2)")]


@pytest.mark.parametrize("name", PENDING)
def test_pending_gpu_suite(name):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, MSAM_RUN_PENDING="1")
    run = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, name), "-m", "gpu", "-q", "--no-header", "-rA"],
                         cwd=os.path.dirname(HERE), env=env, capture_output=True, text=True, timeout=1200)
    tail = "\n".join((run.stdout + "\n" + run.stderr).splitlines()[-40:])
    print(tail)
    assert run.returncode == 0, f"{name}: exit code {run.returncode}\n{tail}"
