"""The reference's own test scripts (competitive_rl/test/test_pong.py, test_car_racing.py), call for call, through this package's
``make_envs`` on the GPU: same arguments, same attribute accesses, the shapes the reference prints.  (test_make_car_racing_double.py builds
ONE gym env through ``make_car_racing_double(...)()``: the single-env facade is not on the vectorised hot path -- the same env as a
batch of one is ``make_envs("cCarRacingDouble-v0", num_envs=1, ...)``, exercised below.)"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")


def test_reference_test_pong_script(tmp_path):
    _need_gpu()
    from competitive_rl_amd import make_envs

    envs = make_envs(
        env_id="cPong-v0",
        seed=0,
        log_dir=str(tmp_path / "demo"),  # this will create a "demo" directory
        num_envs=1,
        asynchronous=False,
        resized_dim=42
    )
    env = envs.envs[0]
    obs = envs.reset()
    env.close()
    assert os.path.isdir(tmp_path / "demo")
    assert tuple(obs.shape) == (1, 4, 42, 42)          # (the reference prints this: FrameStack(4) of (1, 42, 42) frames, one env)
    assert env.observation_space.shape == (4, 42, 42) and env.action_space.n == 3
    envs.close()


def test_reference_test_car_racing_script(tmp_path):
    _need_gpu()
    from competitive_rl_amd import make_envs

    envs = make_envs(
        env_id="cCarRacing-v0",
        seed=0,
        log_dir=str(tmp_path / "demo"),  # this will create a "demo" directory
        num_envs=5,
        asynchronous=True,
        resized_dim=42
    )
    obs = envs.reset()
    assert tuple(obs.shape) == (5, 4, 96, 96)          # frame_stack defaults to 4; resized_dim does not apply to CarRacing (96 x 96 as rendered)
    assert os.path.isdir(tmp_path / "demo")
    envs.close()


def test_reference_make_car_racing_double_as_a_batch_of_one():
    """make_car_racing_double(0, 0, 4, None)(); reset(); step(action_space.sample()); close()  -- as one env of the vector backend"""
    _need_gpu()
    from competitive_rl_amd import make_envs

    e = make_envs("cCarRacingDouble-v0", seed=0, log_dir=None, num_envs=1, frame_stack=4, action_repeat=None)
    obs = e.reset()
    a = e.action_space.sample()                        # Box(-1, 1, (2, 2)): FlattenMultiAgentObservation's action space (utils/atari_wrappers.py:316)
    assert np.asarray(a).shape == (2, 2) and e.action_space.contains(a)
    obs2, rew, done, info = e.step([a])                # one env: a list of one per-env action
    # reward = car 0's (FlattenMultiAgentObservation returns r[0], :329), DummyVecEnv's (N, 1) buffers; both cars' in the infos
    assert tuple(obs.shape) == tuple(obs2.shape) == (1, 8, 96, 96) and tuple(rew.shape) == (1, 1) and tuple(done.shape) == (1, 1)
    assert float(info[0][0]["reward"]) == float(rew[0, 0]) and "reward" in info[0][1]
    obs3, rew3, _, _ = e.step([{0: a[0], 1: a[1]}])    # the raw env's Dict form {car: action} (car_racing_multi_players.py:245) is indexed the same way
    assert tuple(obs3.shape) == tuple(obs.shape)
    e.close()


def test_render_tiles_and_a_vec_env_wrapper_over_the_hip_env():
    """``envs.render("rgb_array")`` of several envs is ONE tiled picture (VecEnv.render -> tile_images, utils/base_vec_env.py:10-38,173-192);
    a user's ``VecEnvWrapper`` subclass (:255-374) wraps what ``make_envs`` returns and finds the env's attributes through the chain."""
    _need_gpu()
    import competitive_rl_amd as crl

    envs = crl.make_envs("cPongDouble-v0", num_envs=3, frame_stack=None, log_dir=None, resized_dim=42)

    class ClipObs(crl.VecEnvWrapper):
        def reset(self):
            return self.venv.reset()

        def step_wait(self):
            o, r, d, i = self.venv.step_wait()
            return o, r * 2, d, i

    w = ClipObs(envs)
    o = w.reset()
    o2, r, d, info = w.step(np.array([[0, 1], [2, 0], [999, 999]]))
    assert tuple(o2[0].shape) == tuple(o[0].shape) == (3, 1, 42, 42) and tuple(r.shape) == (3, 2)
    big = w.render("rgb_array")
    each = envs.get_images()
    assert big.shape == (2 * 210, 2 * 160, 3) and big.dtype == np.uint8
    assert np.array_equal(big[:210, 160:], each[1]) and np.array_equal(big[210:, :160], each[2]) and not big[210:, 160:].any()
    assert w.unwrapped is envs and w.device == envs.device and w.num_envs == 3
    assert crl.make_envs("cPongDouble-v0", num_envs=1, frame_stack=None, log_dir=None).render().shape == (210, 160, 3)
    w.close()


def test_vec_env_constructors_over_thunks_equal_make_envs():
    """The reference builds its batches as ``DummyVecEnv([make_env_a2c_atari(id, seed, i, log_dir, R, K) for i in range(n)])``
    (make_envs.py:100-117; SubprocVecEnv when asynchronous): the same calls here give the batch ``make_envs`` gives -- same observations
    step for step --, a thunk called by itself is env ``rank`` of that batch, and a SubprocVecEnv of ONE env keeps the worker convention."""
    _need_gpu()
    import competitive_rl_amd as crl

    n = 5
    a = crl.DummyVecEnv([crl.make_env_a2c_atari("cPongDouble-v0", 7, i, None, 42, None) for i in range(n)])
    b = crl.make_envs("cPongDouble-v0", seed=7, log_dir=None, num_envs=n, resized_dim=42, frame_stack=None)
    one = crl.make_env_a2c_atari("cPongDouble-v0", 7, 3, None, 42, None)()          # env 3 of the batch, on its own
    sub = crl.SubprocVecEnv([crl.make_env_a2c_atari("cPongDouble-v0", 7, 0, None, 42, None)])
    oa, ob, o1 = a.reset(), b.reset(), one.reset()
    sub.reset()
    rs = np.random.RandomState(0)
    for t in range(300):
        act = rs.randint(0, 3, (n, 2))
        (oa, ra, da, _), (ob, rb, db, _) = a.step(act), b.step(act)
        o1, r1, d1, _ = one.step(act[3:4])
        _, _, ds, _ = sub.step(act[:1])
        assert all(torch.equal(x, y) for x, y in zip(oa, ob)) and torch.equal(ra, rb) and torch.equal(da, db), t
        assert torch.equal(o1[0][0], oa[0][3]) and torch.equal(o1[1][0], oa[1][3]) and torch.equal(r1[0], ra[3]), t
        assert tuple(da.shape) == (n, 2) and tuple(ds.shape) == (1,)
    cars = crl.SubprocVecEnv([crl.make_car_racing_double(1, i, frame_stack=4) for i in range(2)])
    oc = cars.reset()
    oc2, rc, dc, _ = cars.step(np.zeros((2, 2, 2), np.float32))
    assert tuple(oc2.shape) == tuple(oc.shape) == (2, 8, 96, 96) and tuple(dc.shape) == (2,) and tuple(rc.shape) == (2,)
    for e in (a, b, one, sub, cars):
        e.close()


def test_vis_script_runs_a_match_between_builtin_agents(tmp_path):
    """examples/vis.py = the reference's vis.py (minus the window): RULE_BASED against the MEDIUM network for one episode on ONE env."""
    _need_gpu()
    import ast
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "vis.py"), "--left", "RULE_BASED", "--right", "MEDIUM", "-N", "1"],
                         capture_output=True, text=True, timeout=300, cwd=str(tmp_path), env=dict(os.environ, PYTHONPATH=root))
    assert out.returncode == 0, out.stderr[-2000:]
    import re

    last = re.sub(r"np\.float(32|64)\(([^)]*)\)", r"\2", out.stdout.strip().splitlines()[-1])   # (numpy 2 prints its scalars as np.float32(-21.0))
    left, right = ast.literal_eval(last)
    assert sum(left[:3]) == 1 and left[0] == right[2] and left[2] == right[0] and left[3] == -right[3]


def test_reference_car_racing_blackbox_and_action_repetition_scripts():
    """car_racing/test_car_racing.py (``test_blackbox``: 100 sampled actions with ``reset()`` on done; ``test_action_repetition``: the same
    constant action [0.0, 1] for 200 frames as 200 steps of action_repeat=1 and as 40 steps of action_repeat=5) through the single-env
    handle -- the window (``render("human")``) left out.  With action_repeat=5 a step is five physics frames: five times the per-frame
    penalty of a standing start, and the car is further along after the same number of steps."""
    _need_gpu()
    from competitive_rl_amd import make_envs

    e = make_envs("cCarRacing-v0", seed=0, log_dir=None, num_envs=1, frame_stack=None).envs[0]
    o = e.reset()
    assert tuple(o.shape) == (1, 96, 96)
    for _ in range(100):
        o, r, d, info = e.step(e.action_space.sample())
        assert isinstance(d, bool) and tuple(o.shape) == (1, 96, 96)
        if d:
            e.reset()
    e.close()
    rets = {}
    for rep, steps in ((1, 200), (5, 40)):
        envs = make_envs("cCarRacing-v0", seed=0, log_dir=None, num_envs=1, frame_stack=None, action_repeat=rep)
        e = envs.envs[0]
        e.seed(0)
        e.reset()
        total = 0.0
        for _ in range(steps):
            ret = e.step([0.0, 1])
            total += float(ret[1].reshape(-1)[0])
        rets[rep] = (total, int(ret[3]["num_steps"]) if "num_steps" in ret[3] else None, envs.get_state())
        envs.close()
    # 200 physics frames either way (CarRacing.step_count counts frames), full throttle from a standing start on the same track
    assert rets[1][1] == rets[5][1] == 200 or rets[1][1] is None
    assert np.isfinite(rets[1][0]) and np.isfinite(rets[5][0])
