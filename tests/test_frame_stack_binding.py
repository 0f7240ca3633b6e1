"""Host logic of a FrameStackTensor bound to an env (competitive_rl_amd/frame_stack.py), on the CPU: WHEN the stack state an env drew
ahead may replace the reference's update (utils/utils.py:158-170) and when the generic update has to run -- against a plain numpy
restatement of that update, with a small fake env that keeps FrameStackTensor's history rule the way the HIP Pong context does (the
descriptors of the last four planes, erased when an episode ends).  The GPU tests (tests/test_hip_stack_fused.py) run the same
situations through the real env and kernels; this file needs neither."""
import numpy as np
import pytest
import torch

from competitive_rl_amd.frame_stack import FrameStackTensor


class FakeRingEnv:
    """What frame_stack.py uses of HipPongVecEnv, over synthetic one-plane observations: step() / reset() keep the last four planes per
    env (None = erased), draw a bound stack's next state into the buffer it offers, and hand out the newest plane as the observation."""

    def __init__(self, n, shape=(1, 6, 6), seed=0):
        self.n, self.shape, self.rs = n, shape, np.random.RandomState(seed)
        self.device, self.closed = torch.device("cpu"), False
        self._bound_stack = None
        self._serial, self._last_kind, self._learner = 0, None, None
        self.ring = [[None] * 4 for _ in range(n)]
        self._done = torch.zeros(n, dtype=torch.uint8)
        self.draws = 0

    # ---- the hooks
    def _stack_env(self):
        return self

    def _can_draw_stack(self, fst):
        return fst.num_envs == self.n and fst.num_channels == 1 and fst.frame_stack <= 4 and fst.plane_shape == self.shape[1:] and fst.device == self.device

    def _stack_alias(self, fst):
        return False

    def _is_latest_learner_obs(self, obs):
        return self._learner is not None and isinstance(obs, torch.Tensor) and obs.data_ptr() == self._learner.data_ptr() and obs.shape == self._learner.shape

    def _latest_learner_obs(self):
        return self._learner

    def _paint(self, buf, k, valid):
        """planes oldest to newest = ring planes 4 - k .. 3; planes older than `valid` updates are zeros"""
        self.draws += 1
        buf.zero_()
        for i in range(self.n):
            for j in range(k):
                pl = self.ring[i][4 - k + j]
                if pl is not None and j >= k - valid:
                    buf[i, j] = torch.from_numpy(pl[0].astype(np.float32))

    def _draw_stack_into(self, desc):
        # (the descriptor carries the buffer's address, as it does for the library: a host tensor's memory, here)
        import ctypes as C

        count = self.n * desc.planes * int(np.prod(self.shape[1:]))
        flat = np.ctypeslib.as_array((C.c_float * count).from_address(desc.stack_dev))
        buf = torch.from_numpy(flat).view(self.n, desc.planes, *self.shape[1:])
        self._paint(buf, desc.planes, desc.valid_planes)

    def _advance(self, kind, done):
        fst = self._bound_stack() if self._bound_stack is not None else None
        pre = fst._predraw(self, kind) if fst is not None else None
        new = self.rs.randint(0, 256, (self.n, *self.shape)).astype(np.uint8)
        for i in range(self.n):
            if kind == "reset" or done[i]:
                self.ring[i] = [None, None, None, new[i]]   # the history is erased, the new episode's first plane is the newest
            else:
                self.ring[i] = self.ring[i][1:] + [new[i]]
        self._serial += 1
        self._last_kind = kind
        self._done = torch.from_numpy(done.astype(np.uint8))
        self._learner = torch.from_numpy(new.copy())
        if pre is not None:
            buf, desc = pre
            self._paint(buf, desc.planes, desc.valid_planes)
            fst._predrawn(self, buf, kind)
        return self._learner

    def reset(self):
        return self._advance("reset", np.zeros(self.n, bool))

    def step(self, p_done=0.2):
        done = self.rs.random_sample(self.n) < p_done
        return self._advance("step", done), done


class RefStack:
    """utils/utils.py:145-173 in numpy"""

    def __init__(self, n, shape, k):
        self.c, self.buf = shape[0], np.zeros((n, shape[0] * k, *shape[1:]), np.float32)

    def reset(self):
        self.buf[:] = 0

    def update(self, obs, mask=None):
        if mask is not None:
            self.buf *= np.asarray(mask, np.float32).reshape(-1, 1, 1, 1)
        self.buf = np.roll(self.buf, -self.c, axis=1)
        self.buf[:, -self.c:] = np.asarray(obs, np.float32)


def _pair(n=5, k=4, seed=1):
    env = FakeRingEnv(n, seed=seed)
    fst = FrameStackTensor(n, env.shape, k, "cpu")
    # (a host stack has one buffer for its generic update; the bound path swaps two like the device one)
    return env, fst, RefStack(n, env.shape, k)


def _same(fst, ref):
    return np.array_equal(fst.get().numpy(), ref.buf)


def _step_envs_like(env, fst, ref, p_done=0.2):
    obs, done = env.step(p_done)
    mask = (1.0 - done.astype(np.float32)).reshape(-1, 1, 1, 1)
    fst.update(obs, torch.from_numpy(mask), _from_env=env)     # what step_envs passes
    ref.update(obs.numpy(), mask)
    assert _same(fst, ref)


@pytest.mark.parametrize("k", [1, 2, 4])
def test_the_regular_loop_is_all_pointer_swaps(k):
    env, fst, ref = _pair(k=k)
    assert fst.bind(env) and fst._env() is env
    obs = env.reset()
    fst.update(obs), ref.update(obs.numpy())
    assert _same(fst, ref) and fst.fused_updates == 1
    for _ in range(40):
        _step_envs_like(env, fst, ref)
    assert fst.fused_updates == 41


def test_binding_a_used_stack_checks_it_against_the_history_first():
    env, fst, ref = _pair()
    obs = env.reset()
    fst.update(obs), ref.update(obs.numpy())          # unbound: the generic update
    assert fst.fused_updates == 0 and not fst._zero
    draws = env.draws
    assert fst.bind(env) and env.draws == draws + 1 and fst._synced   # one draw + one comparison: the tensor is what the history draws
    for _ in range(6):
        _step_envs_like(env, fst, ref)
    assert fst.fused_updates == 6
    # a stack whose content the env's history canNOT explain binds, stays generic, and comes back once the strange planes have rolled out
    env2, fst2, ref2 = _pair(seed=3)
    env2.reset()
    junk = torch.full((5, 1, 6, 6), 9, dtype=torch.uint8)
    fst2.update(junk), ref2.update(junk.numpy())
    assert fst2.bind(env2) and not fst2._synced
    for _ in range(8):
        _step_envs_like(env2, fst2, ref2, p_done=0.0)
    assert fst2._synced and fst2.fused_updates >= 3


def test_stack_reset_keeps_the_binding_and_draws_only_younger_planes():
    env, fst, ref = _pair()
    fst.bind(env)
    obs = env.reset()
    fst.update(obs), ref.update(obs.numpy())
    for _ in range(5):
        _step_envs_like(env, fst, ref, p_done=0.0)
    fst.reset(), ref.reset()
    assert _same(fst, ref) and fst._spare is None
    for i in range(6):
        _step_envs_like(env, fst, ref, p_done=0.0)
        assert not fst.get()[:, :max(0, 3 - i)].any()
    assert fst.fused_updates == 1 + 5 + 6


def test_what_falls_back_to_the_generic_update_and_how_the_binding_returns():
    env, fst, ref = _pair(n=7)
    fst.bind(env)
    obs = env.reset()
    fst.update(obs), ref.update(obs.numpy())
    for _ in range(5):
        _step_envs_like(env, fst, ref)
    # (1) the env is reset under a live stack, first observation pushed without a mask: the reference keeps the old planes
    obs = env.reset()
    before = fst.fused_updates
    fst.update(obs), ref.update(obs.numpy())
    assert _same(fst, ref) and fst.fused_updates == before and not fst._synced
    for _ in range(6):
        _step_envs_like(env, fst, ref, p_done=0.0)
    assert fst._synced and fst.fused_updates >= before + 2
    # (2) two env steps for one update
    env.step(0.0)
    before = fst.fused_updates
    for _ in range(7):
        _step_envs_like(env, fst, ref, p_done=0.0)
    assert fst._synced and fst.fused_updates >= before + 2
    # (3) a step's observation pushed WITHOUT a mask (the reference then erases nothing, the env's history does on a done): generic
    obs, done = env.step(1.0)
    before = fst.fused_updates
    fst.update(obs), ref.update(obs.numpy())
    assert _same(fst, ref) and fst.fused_updates == before
    # (4) a mask of the caller's own: the stack leaves the env
    obs, done = env.step(0.0)
    m = torch.tensor([1, 0, 1, 1, 0, 1, 1], dtype=torch.float32).reshape(-1, 1, 1, 1)
    fst.update(obs, m), ref.update(obs.numpy(), m.numpy())
    assert _same(fst, ref) and fst._env is None and env._bound_stack is None
    draws = env.draws
    for _ in range(3):
        obs, done = env.step(0.3)
        mask = (1.0 - done.astype(np.float32)).reshape(-1, 1, 1, 1)
        fst.update(obs, torch.from_numpy(mask)), ref.update(obs.numpy(), mask)
        assert _same(fst, ref)
    assert env.draws == draws                       # nothing is drawn ahead for a stack that is not bound
    # (5) update_from_env: the loop of one's own
    assert fst.bind(env)
    for _ in range(8):
        obs, done = env.step(0.0)
        fst.update_from_env(env)
        ref.update(obs.numpy(), (1.0 - done.astype(np.float32)).reshape(-1, 1, 1, 1))
        assert _same(fst, ref)
    assert fst._synced and fst.fused_updates > before


def test_one_env_draws_one_stack_and_a_closed_env_lets_go():
    env, f1, r1 = _pair()
    f2 = FrameStackTensor(5, env.shape, 4, "cpu")
    assert f1.bind(env) and f2.bind(env)
    assert f1._env is None and env._bound_stack() is f2          # the second binding replaces the first
    obs = env.reset()
    f1.update(obs), f2.update(obs), r1.update(obs.numpy())
    assert _same(f1, r1) and _same(f2, r1) and f1.fused_updates == 0 and f2.fused_updates == 1
    env.closed = True
    obs, done = env.step(0.0)
    mask = torch.ones(5, 1, 1, 1)
    f2.update(obs, mask, _from_env=env), r1.update(obs.numpy(), mask.numpy())
    assert _same(f2, r1)                                          # (a closed env: the generic update, no exception)
    assert not FrameStackTensor(5, env.shape, 4, "cpu", out_of_place=False).bind(FakeRingEnv(5))
    assert not FrameStackTensor(5, (2, 6, 6), 2, "cpu").bind(FakeRingEnv(5))     # two channels per observation: not what the env's history holds
    assert not FrameStackTensor(5, env.shape, 4, "cpu").bind(object())
