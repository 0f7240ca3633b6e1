"""bench.py's cpu_baseline leg (oracle/subproc_baseline.py: the C oracle inside the reference's one-process-per-env pipe
architecture, subproc_vec_env.py:11-47) must run for every workload bench.py times -- it broke silently once when the
oracle's render entry point changed."""
import numpy as np
import pytest

from oracle import subproc_baseline as sb


@pytest.mark.parametrize("kind", ["raw", "gray_84", "car"])
def test_subproc_architecture_steps(kind):
    rate, steps, dt = sb.time_subproc(kind, 2, 0.3)
    assert steps >= 1 and rate > 0 and np.isfinite(rate)


@pytest.mark.parametrize("kind", ["raw", "car"])
def test_dummy_architecture_steps(kind):
    rate, steps, dt = sb.time_dummy(kind, 2, 3)
    assert steps == 3 and rate > 0
