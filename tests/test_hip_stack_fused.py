"""FrameStackTensor bound to the HIP Pong env (round 6): ``envs.step`` draws the stack's next state in the launch
that draws the observation (``crl_step_stack``) and ``update`` is a pointer swap.  Every test compares the bound stack with the
generic update sequence of the reference (utils/utils.py:158-170: mask multiply, roll, append) on a twin env with the same seed --
through episode ends, stack resets, env resets under a live stack, foreign updates and re-binds -- at tolerance 0.
``tests/test_step_envs_golden.py`` replays the reference's own recording through the same path."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")


def _twins(n, R, obs_dtype, seed=11):
    import competitive_rl_amd as crl

    mk = lambda: crl.make_envs("cPongDouble-v0", num_envs=n, log_dir=None, seed=seed, resized_dim=R, frame_stack=None, obs_dtype=obs_dtype)
    return mk(), mk()


def _near_the_end(env, which):
    """Puts the listed envs one round before the end of their episode (21 rounds, pong/register.py:20-22)."""
    st = env.get_state()
    st["num_rounds"][which] = 20
    env.set_state(st)


def _books(n, dev):
    return dict(ep=torch.zeros((n, 2), dtype=torch.float32, device=dev), rr=[], lr=[], steps=0, episodes=0)


def _step_envs(crl, env, fst, acts, b, dev):
    out = crl.step_envs(acts, env, b["ep"], fst, b["rr"], b["lr"], b["steps"], b["episodes"], dev, False)
    b["episodes"], b["steps"] = out[5], out[6]
    return out


@pytest.mark.parametrize("obs_dtype", ["uint8", "float32", "float32_ref"])
@pytest.mark.parametrize("R,k", [(84, 4), (42, 4), (84, 2), (42, 1)])
def test_bound_stack_equals_the_update_sequence_through_episode_ends(obs_dtype, R, k):
    _need_gpu()
    import competitive_rl_amd as crl

    n, steps = 96, 60
    a, b = _twins(n, R, obs_dtype)
    dev = a.device
    fa, fb = crl.FrameStackTensor(n, (1, R, R), k, dev), crl.FrameStackTensor(n, (1, R, R), k, dev)
    fb._bind_tried = True  # the twin stays on the generic kernel
    ba, bb = _books(n, dev), _books(n, dev)
    rs = np.random.RandomState(3)
    fa.update(a.reset()[0])
    fb.update(b.reset()[0])
    assert torch.equal(fa.get(), fb.get())
    ends = 0
    for t in range(steps):
        if t in (2, 17, 33, 47):  # four waves of episode ends, a different third of the envs each
            which = np.flatnonzero(rs.random_sample(n) < 0.35)
            _near_the_end(a, which), _near_the_end(b, which)
        acts = torch.as_tensor(rs.randint(0, 3, (n, 2)).astype(np.int32)).to(dev)
        oa = _step_envs(crl, a, fa, acts, ba, dev)
        ob = _step_envs(crl, b, fb, acts, bb, dev)
        assert torch.equal(oa[2], ob[2]) and torch.equal(oa[1], ob[1]), t
        for v in range(2):
            assert oa[0][v].shape == ob[0][v].shape and oa[0][v].dtype == ob[0][v].dtype and torch.equal(oa[0][v], ob[0][v]), (t, v)
        assert fa.get().dtype == torch.float32 and torch.equal(fa.get(), fb.get()), t
        ends += int(oa[2].sum())
    assert ends >= 3 * n // 4, ends                      # (every wave ended episodes)
    assert fa._env is not None and fa.fused_updates == steps, (fa.fused_updates, steps)   # every update was the pointer swap
    assert fb._env is None and fb.fused_updates == 0
    a.close(), b.close()


def test_uint8_stack_is_the_float32_stack_as_bytes():
    _need_gpu()
    import competitive_rl_amd as crl

    n, R, k = 128, 84, 4
    a, b = _twins(n, R, "uint8")
    dev = a.device
    fa, fb = crl.FrameStackTensor(n, (1, R, R), k, dev, dtype=torch.uint8), crl.FrameStackTensor(n, (1, R, R), k, dev)
    fb._bind_tried = True
    ba, bb = _books(n, dev), _books(n, dev)
    rs = np.random.RandomState(4)
    fa.update(a.reset()[0]), fb.update(b.reset()[0])
    for t in range(30):
        if t in (3, 14):
            which = np.flatnonzero(rs.random_sample(n) < 0.4)
            _near_the_end(a, which), _near_the_end(b, which)
        acts = torch.as_tensor(rs.randint(0, 3, (n, 2)).astype(np.int32)).to(dev)
        oa = _step_envs(crl, a, fa, acts, ba, dev)
        _step_envs(crl, b, fb, acts, bb, dev)
        assert fa.get().dtype == torch.uint8 and torch.equal(fa.get().float(), fb.get()), t
        # agent 0's observation IS the stack's newest plane (one element type, frame_stack 1: written once)
        assert oa[0][0].data_ptr() == fa.get()[:, k - 1:].data_ptr()
    assert fa.fused_updates == 30
    a.close(), b.close()


def test_stack_reset_env_reset_and_foreign_updates_fall_back_and_rebind():
    """Everything that is not "the env's newest observation with 1 - done" goes through the generic kernel with the reference's
    result, and the binding resumes once the env's history explains the tensor again."""
    _need_gpu()
    import competitive_rl_amd as crl

    n, R, k = 64, 84, 4
    a, b = _twins(n, R, "uint8")
    dev = a.device
    fa, fb = crl.FrameStackTensor(n, (1, R, R), k, dev), crl.FrameStackTensor(n, (1, R, R), k, dev)
    fb._bind_tried = True
    ba, bb = _books(n, dev), _books(n, dev)
    rs = np.random.RandomState(5)
    acts = lambda: torch.as_tensor(rs.randint(0, 3, (n, 2)).astype(np.int32)).to(dev)

    def both(f):
        ra, rb = f(a, fa, ba), f(b, fb, bb)
        assert torch.equal(fa.get(), fb.get())
        return ra, rb

    both(lambda e, f, _: f.update(e.reset()[0]))
    for _ in range(6):
        x = acts()
        both(lambda e, f, bk: _step_envs(crl, e, f, x, bk, dev))
    assert fa.fused_updates == 6
    # 1. FrameStackTensor.reset() in the middle of episodes: only planes younger than the reset are drawn
    fa.reset(), fb.reset()
    for i in range(6):
        x = acts()
        both(lambda e, f, bk: _step_envs(crl, e, f, x, bk, dev))
        assert fa.get()[:, :max(0, k - 1 - i)].abs().sum() == 0
    assert fa.fused_updates == 12
    # 2. the env is reset under a live stack and the trainer pushes the first observation without a mask: the reference keeps the
    #    old episode's planes (the env's history does not) -> generic until they have rolled out, then bound again
    both(lambda e, f, _: f.update(e.reset()[0]))
    assert fa.fused_updates == 12 and fa.get()[:, 0].abs().sum() > 0
    before = fa.fused_updates
    for _ in range(8):
        x = acts()
        both(lambda e, f, bk: _step_envs(crl, e, f, x, bk, dev))
    assert fa._env is not None and fa.fused_updates >= before + 4, (fa.fused_updates, before)
    # 3. an observation that is not the env's (a trainer injecting a frame): generic, rolls out, bound again
    foreign = torch.full((n, 1, R, R), 7, dtype=torch.uint8, device=dev)
    both(lambda e, f, _: f.update(foreign))
    before = fa.fused_updates
    for _ in range(9):
        x = acts()
        both(lambda e, f, bk: _step_envs(crl, e, f, x, bk, dev))
    assert fa._env is not None and fa.fused_updates >= before + 4
    # 4. two env steps for one update (an observation is skipped)
    x = acts()
    a.step(x), b.step(x)
    before = fa.fused_updates
    for _ in range(9):
        x = acts()
        both(lambda e, f, bk: _step_envs(crl, e, f, x, bk, dev))
    assert fa.fused_updates >= before + 4
    # 5. a mask of the caller's own: the stack leaves the env for good (a draw ahead would be thrown away every step)
    x = acts()
    oa, ob = a.step(x), b.step(x)
    m = (torch.rand((n, 1), device=dev) > 0.5).float()
    fa.update(oa[0][0], m), fb.update(ob[0][0], m)
    assert torch.equal(fa.get(), fb.get()) and fa._env is None
    before = fa.fused_updates
    for _ in range(3):
        x = acts()
        fa._bind_tried = True
        both(lambda e, f, bk: _step_envs(crl, e, f, x, bk, dev))
    assert fa.fused_updates == before
    # ... until it is bound again by hand
    assert fa.bind(a)
    for _ in range(9):
        x = acts()
        both(lambda e, f, bk: _step_envs(crl, e, f, x, bk, dev))
    assert fa.fused_updates >= before + 4
    a.close(), b.close()


def test_a_held_tensor_survives_exactly_one_further_update():
    """Lifetime of what get() / update() return: intact through the next update, recycled by the one after -- bound and generic."""
    _need_gpu()
    import competitive_rl_amd as crl

    n, R, k = 32, 84, 4
    a, b = _twins(n, R, "uint8")
    dev = a.device
    for env, bind in ((a, True), (b, False)):
        f = crl.FrameStackTensor(n, (1, R, R), k, dev)
        f._bind_tried = not bind
        bk = _books(n, dev)
        f.update(env.reset()[0])
        rs = np.random.RandomState(6)
        held = held_copy = None
        for t in range(6):
            out = _step_envs(crl, env, f, torch.as_tensor(rs.randint(0, 3, (n, 2)).astype(np.int32)).to(dev), bk, dev)
            if held is not None:
                assert torch.equal(held, held_copy), (bind, t)   # one further update: unchanged
                assert f.get().data_ptr() != held.data_ptr()
            held, held_copy = f.get(), f.get().clone()
            del out
        assert (f.fused_updates > 0) == bind
        f.reset()
        assert f._spare is None and not f.get().any()             # reset() releases the second buffer
    a.close(), b.close()


def test_update_from_env_with_the_device_step_and_the_tournament_wrapper():
    _need_gpu()
    import competitive_rl_amd as crl

    n, R, k = 48, 42, 4
    a, b = _twins(n, R, "uint8")
    dev = a.device
    fa, fb = crl.FrameStackTensor(n, (1, R, R), k, dev), crl.FrameStackTensor(n, (1, R, R), k, dev)
    assert fa.bind(a)
    a.reset(), b.reset()
    fa.update_from_env(a)
    fb.update(b._latest_learner_obs())
    rs = np.random.RandomState(8)
    for t in range(12):
        if t == 4:
            which = np.arange(0, n, 3)
            _near_the_end(a, which), _near_the_end(b, which)
        x = torch.as_tensor(rs.randint(0, 3, (n, 2)).astype(np.int32)).to(dev)
        bufa, _, _ = a.step_device(x)
        bufb, _, done = b.step_device(x)
        fa.update_from_env(a)
        fb.update(bufb[:, 0], (done == 0).float())
        assert torch.equal(bufa, bufb) and torch.equal(fa.get(), fb.get()), t
    assert fa.fused_updates == 13
    a.close(), b.close()

    tw = crl.make_envs("cPongTournament-v0", num_envs=n, log_dir=None, seed=2)
    tw2 = crl.make_envs("cPongTournament-v0", num_envs=n, log_dir=None, seed=2)
    dev = tw.env.device
    f1, f2 = crl.FrameStackTensor(n, (1, 42, 42), 4, dev), crl.FrameStackTensor(n, (1, 42, 42), 4, dev)
    f2._bind_tried = True
    b1 = dict(ep=torch.zeros((n, 1), dtype=torch.float32, device=dev), rr=[], lr=[], steps=0, episodes=0)
    b2 = dict(ep=torch.zeros((n, 1), dtype=torch.float32, device=dev), rr=[], lr=[], steps=0, episodes=0)
    f1.update(tw.reset()), f2.update(tw2.reset())
    for t in range(10):
        x = torch.as_tensor(rs.randint(0, 3, (n,)).astype(np.int32)).to(dev)
        _step_envs(crl, tw, f1, x, b1, dev), _step_envs(crl, tw2, f2, x, b2, dev)
        assert torch.equal(f1.get(), f2.get()), t
    assert f1.fused_updates == 10
    tw.close(), tw2.close()


def test_what_the_fused_draw_refuses():
    _need_gpu()
    import competitive_rl_amd as crl
    from competitive_rl_amd import _native as N

    L = N.load()
    n, R = 16, 84
    env = crl.HipPongVecEnv(n, mode="wrapped", resized_dim=R, frame_stack=1)
    env.reset()
    buf = torch.zeros((n, 4, R, R), device=env.device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def desc(**kw):
        d = dict(stack_dev=buf.data_ptr(), planes=4, dtype=N.CRL_OBS_F32, agent=0, valid_planes=4, alias_newest=0, reserved=0)
        d.update(kw)
        return N.CrlStackDesc(**d)

    assert L.crl_draw_stack(env._h, None, C.byref(desc()), st) == 0
    for bad in (dict(planes=5), dict(planes=0), dict(agent=2), dict(dtype=7), dict(reserved=1), dict(valid_planes=-1), dict(stack_dev=None),
                dict(alias_newest=1), dict(stack_dev=buf.data_ptr() + 4)):   # (alias: float32 stack over a uint8 observation)
        rc = L.crl_draw_stack(env._h, None, C.byref(desc(**bad)), st)
        assert rc == -1, bad
        assert L.crl_ctx_last_error(env._h)
    env.close()
    # the FrameStack wrapper's history (cPong-v0) and raw contexts keep no FrameStackTensor history
    rep = crl.make_envs("cPong-v0", num_envs=n, log_dir=None, resized_dim=R, frame_stack=4)
    rep.reset()
    assert L.crl_draw_stack(rep._h, None, C.byref(desc()), st) == -4
    f = crl.FrameStackTensor(n, (4, R, R), 1, rep.device)
    assert not f.bind(rep)
    rep.close()
    raw = crl.HipPongVecEnv(n, mode="raw")
    raw.reset()
    assert L.crl_draw_stack(raw._h, None, C.byref(desc()), st) == -4
    raw.close()
    # a float32 context writes float32 stacks only; more than four planes, other shapes, in-place stacks stay generic
    f32 = crl.HipPongVecEnv(n, mode="wrapped", resized_dim=R, frame_stack=1, obs_dtype="float32")
    f32.reset()
    assert L.crl_draw_stack(f32._h, None, C.byref(desc(dtype=N.CRL_OBS_U8)), st) == -1
    assert not crl.FrameStackTensor(n, (1, R, R), 4, f32.device, dtype=torch.uint8).bind(f32)
    assert not crl.FrameStackTensor(n, (1, R, R), 5, f32.device).bind(f32)
    assert not crl.FrameStackTensor(n, (1, 42, 42), 4, f32.device).bind(f32)
    assert not crl.FrameStackTensor(n, (1, R, R), 4, f32.device, out_of_place=False).bind(f32)
    assert crl.FrameStackTensor(n, (1, R, R), 4, f32.device).bind(f32)
    f32.close()


def test_full_size_bound_stack_against_the_generic_one():
    """BASELINE config sizes: 65 536 envs, (4, 84, 84) float32 -- 7.4 GB per buffer; eight steps, both stacks compared on the device."""
    _need_gpu()
    import competitive_rl_amd as crl

    n, R, k = 65536, 84, 4
    a, b = _twins(n, R, "uint8", seed=0)
    dev = a.device
    fa, fb = crl.FrameStackTensor(n, (1, R, R), k, dev), crl.FrameStackTensor(n, (1, R, R), k, dev)
    fb._bind_tried = True
    ba, bb = _books(n, dev), _books(n, dev)
    fa.update(a.reset()[0]), fb.update(b.reset()[0])
    st = a.get_state()
    st["num_rounds"][::5] = 20
    a.set_state(st), b.set_state(st)
    gen = torch.Generator(device=dev).manual_seed(1)
    for t in range(8):
        acts = torch.randint(0, 3, (n, 2), generator=gen, device=dev, dtype=torch.int32)
        _step_envs(crl, a, fa, acts, ba, dev), _step_envs(crl, b, fb, acts, bb, dev)
        assert torch.equal(fa.get(), fb.get()), t
    assert fa.fused_updates == 8 and ba["episodes"] == bb["episodes"] and ba["episodes"] > 0
    a.close(), b.close()


def test_done_host_hands_over_the_flags_of_the_last_step():
    """crl_set_flags_event: the host copy of the done flags is ordered behind the dynamics kernel only, and is what step() returned."""
    _need_gpu()
    import competitive_rl_amd as crl

    n = 4096
    env = crl.make_envs("cPongDouble-v0", num_envs=n, log_dir=None, seed=3, resized_dim=84, frame_stack=None)
    env.reset()
    with pytest.raises(RuntimeError):
        env.done_host()
    rs = np.random.RandomState(9)
    seen = 0
    for t in range(40):
        if t % 7 == 3:
            _near_the_end(env, np.flatnonzero(rs.random_sample(n) < 0.3))
        x = torch.as_tensor(rs.randint(0, 3, (n, 2)).astype(np.int32)).to(env.device)
        if t % 2:
            _, _, done, _ = env.step(x)
            done = done[:, 0]
        else:
            _, _, done = env.step_device(x)
        h = env.done_host()
        assert h.dtype == bool and np.array_equal(h, done.cpu().numpy().astype(bool)), t
        assert env.done_host() is not None  # (a second call is served from the same copy)
        seen += int(h.sum())
    assert seen > n // 4
    env.close()


@pytest.mark.parametrize("obs_dtype,stack_dtype", [("uint8", torch.float32), ("uint8", torch.uint8), ("float32_ref", torch.float32)])
def test_bound_stack_long_run_with_natural_episode_ends(obs_dtype, stack_dtype):
    """1 500 steps x 2 048 envs of random play: episodes end on their own (21 rounds), at every score pair a game passes through, with
    points scored between the two kept frames of a plane (the score-transition tables) -- the bound stack against the generic one, every step."""
    _need_gpu()
    import competitive_rl_amd as crl

    n, R, k, steps = 2048, 84, 4, 1500
    a, b = _twins(n, R, obs_dtype, seed=23)
    dev = a.device
    fa, fb = crl.FrameStackTensor(n, (1, R, R), k, dev, dtype=stack_dtype), crl.FrameStackTensor(n, (1, R, R), k, dev)
    fb._bind_tried = True
    ba, bb = _books(n, dev), _books(n, dev)
    fa.update(a.reset()[0]), fb.update(b.reset()[0])
    gen = torch.Generator(device=dev).manual_seed(4)
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    for t in range(steps):
        acts = torch.randint(0, 3, (n, 2), generator=gen, device=dev, dtype=torch.int32)
        _step_envs(crl, a, fa, acts, ba, dev), _step_envs(crl, b, fb, acts, bb, dev)
        bad += (fa.get().float() != fb.get()).any().to(torch.int64)   # (no host synchronisation per step: summed on the device)
    assert int(bad) == 0, int(bad)
    assert fa.fused_updates == steps and ba["episodes"] == bb["episodes"] and ba["episodes"] > n, (fa.fused_updates, ba["episodes"])
    st = a.get_state()
    assert st["score_l"].max() >= 10 and st["score_r"].max() >= 10   # (late-game score pairs were on the screen)
    a.close(), b.close()
