"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle and the
golden vectors recorded from the reference.  Bit-exact everywhere (integer / byte work;
the f64 ball speeds are compared as bit patterns)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")


def rand_actions(rs, steps, n, cheat=0.15):
    a = rs.randint(0, 3, (steps, n, 2)).astype(np.int32)
    a[rs.random_sample((steps, n, 2)) < cheat] = 999
    return a


STATE_FIELDS = ["speed_x", "speed_y", "ball_x", "ball_y", "bat_l_y", "bat_r_y", "score_l", "score_r",
                "num_rounds", "num_steps", "serve_ctr", "wrap_steps"]


def assert_state_equal(hs, os_, fields=STATE_FIELDS, ctx=""):
    for f in fields:
        a, b = hs[f], os_[f]
        if a.dtype.kind == "f":
            a, b = a.view(np.uint64), b.view(np.uint64)
        bad = np.nonzero(a != b)[0]
        assert bad.size == 0, f"{ctx}: field {f} differs at envs {bad[:8]}: hip={hs[f][bad[:4]]} oracle={os_[f][bad[:4]]}"


@pytest.mark.parametrize("name", ["random_a", "random_cheat", "rule_vs_rule", "sticky", "timeout"])
def test_golden_traces_through_hip(golden_dyn, atlas, name):
    """The reference's own PongGame traces replayed on the GPU (replay-mode serves)."""
    _need_gpu()
    import competitive_rl_amd as crl

    g = {k.split("/", 1)[1]: golden_dyn[k] for k in golden_dyn.files if k.startswith(name + "/")}
    T = len(g["acts"])
    env = crl.HipPongVecEnv(1, mode="raw")
    env.set_replay(g["draw_u"][None], g["draw_bx"][None], g["draw_by"][None])
    env.reset()
    inject = dict(zip(g["inject_t"].tolist(), g["inject_v"].tolist()))
    acts = torch.as_tensor(g["acts"][:, None, :].astype(np.int32)).cuda()
    rews = torch.zeros((T, 2), device="cuda")
    dones = torch.zeros((T,), dtype=torch.uint8, device="cuda")
    F = {n: i for i, n in enumerate(g.get("fields", golden_dyn["fields"]).tolist())}
    check_at = set(range(0, T, 250)) | {T - 1}
    for t in range(T):
        if t in inject:
            st = env.get_state()
            st["num_steps"][0] = inject[t]
            env.set_state(st)
        _, r, d = env.step_device(acts[t], render=False)
        rews[t], dones[t] = r[0], d[0]
        if t in check_at:
            st = env.get_state()[0]
            post = g["post"][t]
            i64 = post.view(np.int64)
            got = (st["ball_x"], st["ball_y"], np.float64(st["speed_x"]).view(np.uint64),
                   np.float64(st["speed_y"]).view(np.uint64), st["bat_l_y"], st["bat_r_y"], st["score_l"],
                   st["score_r"], st["num_rounds"], st["num_steps"])
            want = (i64[F["ball_x"]], i64[F["ball_y"]], post[F["sx_bits"]], post[F["sy_bits"]], i64[F["bat_l"]],
                    i64[F["bat_r"]], i64[F["score_l"]], i64[F["score_r"]], i64[F["rounds"]], i64[F["steps"]])
            assert tuple(int(x) for x in got) == tuple(int(x) for x in want), (name, t)
            assert int(st["serve_ctr"]) == int(g["ndraws"][t])
    assert np.array_equal(rews.cpu().numpy().astype(np.int32), g["rew"])
    assert np.array_equal(dones.cpu().numpy(), g["done"])
    env.close()


def test_raw_step_matches_oracle_bit_exact(atlas):
    """Dynamics + raw (N,2,210,160,3) raster vs the oracle, Philox serves, incl. action 999."""
    _need_gpu()
    import competitive_rl_amd as crl
    from oracle import pong_oracle as po

    n, steps = 192, 700
    rs = np.random.RandomState(11)
    acts = rand_actions(rs, steps, n)
    env = crl.HipPongVecEnv(n, seed=5, mode="raw")
    ora = po.PongOracle(n, atlas, obs_mode=po.RAW, seed=5)
    o_h = env.reset()
    o_o = ora.reset()
    assert np.array_equal(torch.stack(o_h, 1).cpu().numpy(), o_o)
    n_done = 0
    for t in range(steps):
        render = (t % 23 == 0) or t > steps - 4
        obs, rew, done, infos = env.step(acts[t])
        oo, orew, odone = ora.step(acts[t], render=render)
        assert np.array_equal(rew.cpu().numpy(), orew), t
        assert np.array_equal(done[:, 0].cpu().numpy().astype(np.uint8), odone), t
        if render:
            assert np.array_equal(torch.stack(obs, 1).cpu().numpy(), oo), t
        for i in np.nonzero(odone)[0][:2]:
            n_done += 1
            term = infos[int(i)]["terminal_observation"]
            want = ora.terminal_observation(int(i))
            assert np.array_equal(torch.stack(term).cpu().numpy(), want)
        if t % 100 == 0:
            assert_state_equal(env.get_state(), ora.state, ctx=f"t={t}")
    assert_state_equal(env.get_state(), ora.state, ctx="final")
    assert n_done > 0
    env.close()


@pytest.mark.parametrize("R,K", [(84, 1), (84, 4), (42, 1), (42, 4)])
def test_wrapped_step_matches_oracle_bit_exact(atlas, R, K):
    """skip-4 / max-2 / gray / INTER_AREA / stack fused kernel vs the oracle's literal pipeline."""
    _need_gpu()
    import competitive_rl_amd as crl
    from oracle import pong_oracle as po

    n, steps = 96, 260
    rs = np.random.RandomState(100 + R + K)
    acts = rand_actions(rs, steps, n)
    env = crl.HipPongVecEnv(n, seed=9, mode="wrapped", resized_dim=R, frame_stack=K)
    ora = po.PongOracle(n, atlas, obs_mode=po.GRAY, resized_dim=R, frame_stack=K, seed=9)
    o_h = torch.stack(env.reset(), 1).cpu().numpy()
    o_o = ora.reset()
    assert o_h.shape == o_o.shape == (n, 2, K, R, R)
    assert np.array_equal(o_h, o_o)
    n_done = n_term = 0
    for t in range(steps):
        obs, rew, done, infos = env.step(acts[t])
        oo, orew, odone = ora.step(acts[t])
        assert np.array_equal(rew.cpu().numpy(), orew), t
        assert np.array_equal(done[:, 0].cpu().numpy().astype(np.uint8), odone), t
        got = torch.stack(obs, 1).cpu().numpy()
        bad = np.argwhere(got != oo)
        assert bad.size == 0, (t, bad[:5], got[tuple(bad[0])], oo[tuple(bad[0])])
        i0 = infos[0]
        assert i0["real_reward"] == [float(ora.real_reward[0, 0]), float(ora.real_reward[0, 1])]
        assert i0["num_steps"] == int(ora.num_steps[0])
        for i in np.nonzero(odone)[0][:2]:
            n_done += 1
            term = infos[int(i)]["terminal_observation"]
            want = ora.terminal_observation(int(i))
            assert term[0].shape == (1, R, R)
            assert np.array_equal(torch.stack(term).cpu().numpy()[:, 0], want)
            n_term += 1
    assert_state_equal(env.get_state(), ora.state, ctx="final")
    hs, os_ = env.get_state(), ora.state
    assert np.array_equal(hs["keep"], os_["keep"])
    assert np.array_equal(hs["hist"], os_["hist"])
    assert n_done > 0
    env.close()


def test_score_change_between_kept_frames_slow_path(atlas):
    """A point scored between the two max-pooled frames puts two score texts under the max:
    the kernel's per-pixel slow path must agree with the oracle."""
    _need_gpu()
    import competitive_rl_amd as crl
    from oracle import pong_oracle as po

    n = 64
    env = crl.HipPongVecEnv(n, seed=1, mode="wrapped", resized_dim=84, frame_stack=4)
    ora = po.PongOracle(n, atlas, obs_mode=po.GRAY, resized_dim=84, frame_stack=4, seed=1)
    env.reset(), ora.reset()
    # put every ball one frame away from leaving on the left, staggered so that the point
    # lands on frame index 0..3 of the wrapped step
    st = ora.state
    for i in range(n):
        st["ball_x"][i] = 1 + 4 * (i % 4)
        st["ball_y"][i] = 60 + i
        st["speed_x"][i] = -4.0
        st["speed_y"][i] = 0.5
        st["score_l"][i], st["score_r"][i] = i % 7, (i // 7) % 9
    env.set_state(np.array(st))
    acts = np.ones((n, 2), np.int32)
    hits = 0
    for t in range(3):
        obs, rew, done, _ = env.step(acts)
        oo, orew, _ = ora.step(acts)
        assert np.array_equal(rew.cpu().numpy(), orew)
        assert np.array_equal(torch.stack(obs, 1).cpu().numpy(), oo), t
        hits += int((orew != 0).any(axis=1).sum())
    k = ora.state["keep"]
    assert hits >= n // 2
    env.close()


def test_sharding_invariance(atlas):
    """Two shards with env_id_base = 0 / n reproduce one 2n batch (RNG keyed by global id)."""
    _need_gpu()
    import competitive_rl_amd as crl

    n, steps = 128, 300
    rs = np.random.RandomState(3)
    acts = rand_actions(rs, steps, 2 * n)
    big = crl.HipPongVecEnv(2 * n, seed=77, mode="wrapped", resized_dim=84)
    s0 = crl.HipPongVecEnv(n, seed=77, mode="wrapped", resized_dim=84, env_id_base=0)
    s1 = crl.HipPongVecEnv(n, seed=77, mode="wrapped", resized_dim=84, env_id_base=n)
    for e in (big, s0, s1):
        e.reset()
    for t in range(steps):
        ob, rb, db, _ = big.step(acts[t])
        o0, r0, d0, _ = s0.step(acts[t, :n])
        o1, r1, d1, _ = s1.step(acts[t, n:])
        assert torch.equal(rb, torch.cat([r0, r1])) and torch.equal(db, torch.cat([d0, d1]))
        assert torch.equal(ob[0], torch.cat([o0[0], o1[0]])) and torch.equal(ob[1], torch.cat([o0[1], o1[1]]))
    for e in (big, s0, s1):
        e.close()


def test_full_size_raw_properties():
    """BASELINE config #2 at its real size (65 536 envs): size-independent invariants."""
    _need_gpu()
    import competitive_rl_amd as crl

    n = 65536
    env = crl.HipPongVecEnv(n, seed=0, mode="raw")
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    for t in range(40):
        a = torch.randint(0, 3, (n, 2), generator=g, device="cuda", dtype=torch.int32)
        buf, rew, done = env.step_device(a)
    torch.cuda.synchronize()
    v0, v1 = buf[:, 0], buf[:, 1]
    # second agent's view: rows < 25 identical, rows >= 25 mirrored (base_pong_env.py:153-154)
    assert torch.equal(v1[:, :25], v0[:, :25])
    assert torch.equal(v1[:, 25:], v0[:, 25:].flip(2))
    # achromatic, white borders, arena holds only 0/255
    assert torch.equal(v0[..., 0], v0[..., 1]) and torch.equal(v0[..., 0], v0[..., 2])
    assert bool((v0[:, 194:] == 255).all()) and bool((v0[:, :8] == 255).all())
    arena = v0[:, 34:194, :, 0]
    assert bool(((arena == 0) | (arena == 255)).all())
    # white arena pixels = ball (<=16) + two bats (75 each); only the ball can be clipped/overlap
    white = (arena == 255).sum(dim=(1, 2))
    assert int(white.max()) <= 166 and int(white.min()) >= 150
    # zero-sum rewards
    assert bool((rew.sum(dim=1) == 0).all())
    # determinism: same seed, same actions -> same bytes (checksum of checksums)
    ck1 = int(buf.reshape(-1).view(torch.int32).sum(dtype=torch.int64))
    env2 = crl.HipPongVecEnv(n, seed=0, mode="raw")
    env2.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    for t in range(40):
        a = torch.randint(0, 3, (n, 2), generator=g, device="cuda", dtype=torch.int32)
        buf2, _, _ = env2.step_device(a)
    assert int(buf2.reshape(-1).view(torch.int32).sum(dtype=torch.int64)) == ck1
    st1, st2 = env.get_state(), env2.get_state()
    assert_state_equal(st1, st2, ctx="determinism")
    env.close(), env2.close()


def test_full_size_wrapped_properties():
    """BASELINE config #3 at its real size: (65 536, 2, 4, 84, 84) fused output."""
    _need_gpu()
    import competitive_rl_amd as crl

    n = 65536
    env = crl.HipPongVecEnv(n, seed=0, mode="wrapped", resized_dim=84, frame_stack=4)
    first = env.reset()
    # after reset the stack is [0, 0, 0, obs]
    assert bool((first[0][:, :3] == 0).all()) and bool((first[0][:, 3] != 0).any())
    g = torch.Generator(device="cuda").manual_seed(1)
    prev = None
    for t in range(12):
        a = torch.randint(0, 3, (n, 2), generator=g, device="cuda", dtype=torch.int32)
        buf, rew, done = env.step_device(a)
        if prev is not None:
            keep = ~done.bool()
            # FrameStackTensor roll: plane k of this step == plane k+1 of the previous step
            assert torch.equal(buf[keep][:, :, :3], prev[keep][:, :, 1:])
        prev = buf.clone()
    # bottom rows of every non-blank plane are the white band; top-left is white too
    newest = buf[:, :, 3]
    assert bool((newest[:, :, 79:, :] == 255).all()) and bool((newest[:, :, :3, :6] == 255).all())
    env.close()


def test_make_envs_api_surface():
    """Shapes / dtypes / info keys of the reference's DummyVecEnv contract."""
    _need_gpu()
    import competitive_rl_amd as crl

    envs = crl.make_envs("cPongDouble-v0", num_envs=4, asynchronous=False, frame_stack=None, log_dir=None,
                         output="numpy", obs_dtype="float32")
    obs = envs.reset()
    assert isinstance(obs, tuple) and len(obs) == 2 and obs[0].shape == (4, 1, 42, 42) and obs[0].dtype == np.float32
    acts = np.random.RandomState(0).randint(0, 3, (1000, 4, 2))
    for t in range(200):
        o, r, d, info = envs.step(acts[t])
    assert r.shape == (4, 2) and r.dtype == np.float32 and set(np.unique(r)) <= {-1.0, 0.0, 1.0}
    assert d.shape == (4, 2) and d.dtype == np.bool_
    assert set(info[0].keys()) >= {"real_reward", "num_steps"} and len(info) == 4
    assert envs.observation_space[0].shape == (1, 42, 42) and envs.action_space[0].n == 3
    envs.envs[0].close()
    envs.close()
    envs.close()  # idempotent
    sub = crl.make_envs("cPongDouble-v0", num_envs=3, asynchronous=True, frame_stack=None, log_dir=None)
    sub.reset()
    _, _, d, _ = sub.step([[0, 0], [1, 0], [2, 1]])
    assert tuple(d.shape) == (3,)
    with pytest.raises(AssertionError):
        sub.step([[0, 0]])
    with pytest.raises(AssertionError):
        crl.make_envs("cPongDouble-v0", num_envs=2, log_dir=None)  # frame_stack default 4 is rejected
    sub.close()


def test_reference_wrapper_golden_through_hip():
    """BASELINE config #1 recorded from the reference's wrappers + DummyVecEnv
    (tests/golden/pong_wrapped.npz), replayed on the GPU: rewards, dones, infos, obs,
    terminal observations."""
    _need_gpu()
    import os

    import competitive_rl_amd as crl

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pong_wrapped.npz"))
    N, R = g["acts"].shape[1], int(g["resized_dim"])
    blank = np.full((22, 22, 34, 160), 255, np.uint8)  # golden frames carry no score text
    env = crl.HipPongVecEnv(N, mode="wrapped", resized_dim=R, frame_stack=1, score_atlas=blank, output="numpy",
                            obs_dtype="float32")
    env.set_replay(g["draw_u"], g["draw_bx"], g["draw_by"])
    o = env.reset()
    assert o[0].dtype == np.float32 and o[0].shape == (N, 1, R, R)
    assert np.array_equal(np.stack([o[0][:, 0], o[1][:, 0]], 1), g["obs0"])
    term = {(int(t), int(i)): k for k, (t, i) in enumerate(zip(g["term_t"], g["term_i"]))}
    seen = 0
    for t in range(len(g["acts"])):
        o, r, d, info = env.step(g["acts"][t])
        assert np.array_equal(r, g["rew"][t]) and np.array_equal(d, g["done"][t]), t
        assert np.array_equal(np.stack([o[0][:, 0], o[1][:, 0]], 1), g["obs"][t]), t
        if t % 50 == 0 or d.any():
            for i in range(N):
                assert info[i]["real_reward"] == g["real_reward"][t, i].tolist()
                assert info[i]["num_steps"] == int(g["num_steps"][t, i])
                if d[i, 0]:
                    to = info[i]["terminal_observation"]
                    assert np.array_equal(np.stack([to[0][0], to[1][0]]), g["term_obs"][term[(t, i)]])
                    seen += 1
    assert seen == len(g["term_t"])
    env.close()


def test_single_player_golden_and_tournament_shapes():
    """cPong-v0 (PongSinglePlayerEnv + FrameStack 4) recorded from the reference, replayed on the
    GPU; then the make_envs.py:121-170 self-test: cPong-v0 and cPongTournament-v0 agree on shapes."""
    _need_gpu()
    import os

    import competitive_rl_amd as crl

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pong_single_wrapped.npz"))
    N, R, K = g["acts"].shape[1], int(g["resized_dim"]), int(g["frame_stack"])
    blank = np.full((22, 22, 34, 160), 255, np.uint8)
    env = crl.HipPongVecEnv(N, mode="wrapped", resized_dim=R, frame_stack=K, score_atlas=blank, output="numpy",
                            single_player=True, stack_replicate=True)
    env.set_replay(g["draw_u"], g["draw_bx"], g["draw_by"])
    o = env.reset()
    assert o.shape == (N, K, R, R) and np.array_equal(o, g["obs0"])
    term = {(int(t), int(i)): k for k, (t, i) in enumerate(zip(g["term_t"], g["term_i"]))}
    seen = 0
    for t in range(len(g["acts"])):
        o, r, d, info = env.step(g["acts"][t])
        assert r.shape == (N, 1) and d.shape == (N, 1)
        assert np.array_equal(r, g["rew"][t]) and np.array_equal(d, g["done"][t]), t
        assert np.array_equal(o, g["obs"][t]), t
        for i in np.nonzero(d[:, 0])[0]:
            inf = info[int(i)]
            assert inf["real_reward"] == float(g["real_reward"][t, i]) and inf["num_steps"] == int(g["num_steps"][t, i])
            assert np.array_equal(inf["terminal_observation"], g["term_obs"][term[(t, int(i))]])
            seen += 1
    assert seen == len(g["term_t"])
    env.close()

    envs = crl.make_envs("cPong-v0", num_envs=3, log_dir=None, asynchronous=False)
    tour = crl.make_envs("cPongTournament-v0", num_envs=3, log_dir=None, asynchronous=False)
    assert tuple(envs.reset().shape) == (3, 4, 42, 42)  # frame_stack = 4 default
    assert tuple(tour.reset().shape) == (3, 1, 42, 42)
    o1, r1, d1, _ = envs.step([0, 1, 0])
    o2, r2, d2, _ = tour.step([0, 1, 0])
    assert tuple(r1.shape) == tuple(r2.shape) == (3, 1) and tuple(d1.shape) == tuple(d2.shape) == (3, 1)
    assert envs.action_space.n == 3 and tour.action_space.n == 3
    # RULE_BASED opponent == the AutoBat of cPong-v0: same game when stacks are ignored
    e1 = crl.make_envs("cPong-v0", num_envs=5, log_dir=None, frame_stack=None, seed=4)
    e2 = crl.make_envs("cPongTournament-v0", num_envs=5, log_dir=None, seed=4)
    a, b = e1.reset(), e2.reset()
    assert torch.equal(a, b)
    rs = np.random.RandomState(0)
    for t in range(150):
        act = rs.randint(0, 3, 5)
        a, ra, da, _ = e1.step(act)
        b, rb, db, _ = e2.step(act)
        assert torch.equal(a, b) and torch.equal(ra, rb) and torch.equal(da, db)
    for e in (envs, tour, e1, e2):
        e.close()


@pytest.mark.parametrize("n", [1, 3, 41])
def test_raw_single_player_and_ragged_sizes(atlas, n):
    """cPong-v0 raw frames (one view per env: the VIEWS = 1 instance of the address-linear raster)
    and env counts whose chunk totals are not multiples of the 512-chunk workgroup span."""
    _need_gpu()
    import competitive_rl_amd as crl
    from oracle import pong_oracle as po

    steps = 260
    rs = np.random.RandomState(100 + n)
    acts = rs.randint(0, 3, (steps, n)).astype(np.int32)
    env = crl.HipPongVecEnv(n, seed=9, mode="raw", single_player=True)
    ora = po.PongOracle(n, atlas, obs_mode=po.RAW, seed=9, single=True)
    o_h, o_o = env.reset(), ora.reset()
    assert np.array_equal(o_h.cpu().numpy().reshape(o_o.shape), o_o)
    for t in range(steps):
        obs, rew, done, infos = env.step(acts[t])
        oo, orew, odone = ora.step(acts[t], render=True)
        assert np.array_equal(rew.cpu().numpy().reshape(orew.shape), orew), t
        assert np.array_equal(obs.cpu().numpy().reshape(oo.shape), oo), t
    env.close()
    # double-player raw at the same ragged sizes
    env = crl.HipPongVecEnv(n, seed=9, mode="raw")
    ora = po.PongOracle(n, atlas, obs_mode=po.RAW, seed=9)
    o_h, o_o = env.reset(), ora.reset()
    assert np.array_equal(torch.stack(o_h, 1).cpu().numpy(), o_o)
    acts2 = rs.randint(0, 3, (60, n, 2)).astype(np.int32)
    for t in range(60):
        obs, rew, done, infos = env.step(acts2[t])
        oo, orew, odone = ora.step(acts2[t], render=True)
        assert np.array_equal(torch.stack(obs, 1).cpu().numpy(), oo), t
    env.close()


def test_c_abi_demo_runs_without_python_in_the_loop():
    """The standalone C++ caller of include/crl.h: steps 2 048 envs through crl_create / crl_reset /
    crl_step / crl_get_state and checks zero-sum rewards and a plausible frame itself (exit code); then the frame stack drawn by
    the step (crl_step_stack) against the rolled-and-appended one (crl_frame_stack_update_to) on 512 envs through episode ends."""
    _need_gpu()
    import subprocess

    from competitive_rl_amd.build import build_c_demo

    exe = build_c_demo()
    r = subprocess.run([exe, "2048", "120"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "env-steps/s" in r.stdout
    # round 6: the demo also runs the trainer's float32 frame stack through crl_step_stack / crl_draw_stack / crl_set_flags_event from plain C
    assert "0 of " in r.stdout and "elements differ" in r.stdout, r.stdout


def test_step_envs_device_and_host_flows_agree():
    """step_envs (utils/utils.py:23-60): the reference's host flow (numpy env outputs, numpy episode_rewards) and the
    device flow (torch env outputs, device episode_rewards, device FrameStackTensor) keep identical books."""
    _need_gpu()
    import competitive_rl_amd as crl

    n, T = 24, 600
    outs = {}
    for mode in ("numpy", "torch"):
        env = crl.make_envs("cPongDouble-v0", num_envs=n, frame_stack=None, log_dir=None, seed=8, output=mode)
        obs = env.reset()
        fst = crl.FrameStackTensor(n, (1, 42, 42), 4, "cuda")
        fst.update(obs[0])
        ep = np.zeros((n, 1), np.float32) if mode == "numpy" else torch.zeros((n, 1), device="cuda")
        rr, lr, steps, eps = [], [], 0, 0
        rs = np.random.RandomState(4)
        for t in range(T):
            acts = np.stack([rs.randint(0, 3, n), np.full(n, 999)], 1)
            env_step = env.step

            class OneAgentReward:  # the trainers use single-agent rewards: feed agent 0's column
                num_envs = n

                @staticmethod
                def step(a):
                    o, r, d, i = env_step(a)
                    return o, r[:, :1], d, i

            _, _, done, _, masks, eps, steps, ep = crl.step_envs(acts, OneAgentReward, ep, fst, rr, lr, steps, eps, "cuda", False)
            assert tuple(masks.shape) == (n, 1) and tuple(done.shape) == (n,)
        outs[mode] = (np.array([float(np.asarray(x).reshape(-1)[0]) for x in rr]), list(lr), steps, eps,
                      fst.get().cpu().numpy(), ep.cpu().numpy() if mode == "torch" else ep)
        env.close()
    a, b = outs["numpy"], outs["torch"]
    assert a[3] == b[3] > 0 and a[2] == b[2] == n * T
    assert np.array_equal(a[0], b[0]) and a[1] == b[1]
    assert np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])
