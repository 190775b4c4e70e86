"""The harness row (SURVEY.md 8 a-H): every script under examples/ runs on the engine at a small size (the flows of
examples/demo_model_comparison.py:61-74, demo_gaussian_er_gibbs.py:36-43, demo_ais.py:10-62 of the reference)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("demo_gaussian_er_gibbs.py", ["1234", "30", "2", "8"]), ("demo_gaussian_sbm_gibbs.py", ["96", "3", "6"]),
         ("demo_joint_sbm_gibbs.py", ["90", "3", "6"]), ("demo_model_comparison.py", ["20", "6"]),
         ("demo_eigenmodel_gibbs.py", ["logistic", "20", "2", "8"]), ("demo_eigenmodel_gibbs.py", ["probit", "20", "2", "8"]),
         ("demo_distance_model_gibbs.py", ["20", "8"]), ("demo_gaussian_distance_model.py", ["20", "8"]), ("demo_ais.py", ["24", "2", "8"])]


@pytest.mark.gpu
@pytest.mark.parametrize("script,args", CASES)
def test_example_runs(script, args):
    env = dict(os.environ, GG_EXAMPLE_SMALL="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script)] + args, cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-800:], out.stderr[-2000:])
    assert len(out.stdout) > 0
