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


@pytest.mark.parametrize("full", [False, True])
def test_tournament_baseline_leg_runs(full):
    """bench.py's CPU restatement of the tournament loop (oracle Pong env + numpy opponent network), light and full-size."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from oracle import pong_oracle as po

    atlas = np.load(os.path.join(root, "competitive_rl_amd", "assets", "pong_score_atlas.npz"))["atlas"]
    b = bench.cpu_baseline_tournament(po, atlas, 2, 0.5, full=full)
    assert b["value"] > 0 and b["kind"] == "port"
