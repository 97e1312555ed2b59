"""The example scripts (upstream notebooks with the imports switched) run end to end on the GPU."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["run_cahn_hilliard", "thomas_fermi", "pde_env_random_policy", "smoothed_boundary", "gpe_stirring_control"])
def test_example_runs(name):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", name + ".py"), "--quick"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
