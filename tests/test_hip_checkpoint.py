"""Checkpoint / resume: ``state_dict()`` / ``load_state_dict()`` of the two env families -- a run resumed from a
checkpoint produces the bytes the original run produced."""
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")


def _roundtrip(sd):
    """through torch.save / torch.load, as a trainer's checkpoint would go"""
    buf = io.BytesIO()
    torch.save(sd, buf)
    buf.seek(0)
    return torch.load(buf, weights_only=False)


def test_pong_resume_from_state_dict_reproduces_the_run():
    _need_gpu()
    import competitive_rl_amd as crl

    n = 256
    env = crl.HipPongVecEnv(n, seed=3, mode="wrapped", resized_dim=42, frame_stack=4)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    acts = [torch.randint(0, 3, (n, 2), generator=g, device="cuda", dtype=torch.int32) for _ in range(60)]
    for a in acts[:20]:
        env.step_device(a)
    sd = _roundtrip(env.state_dict())
    assert sd["kind"] == "cPong" and sd["num_envs"] == n
    want = [tuple(t.clone() for t in env.step_device(a)) for a in acts[20:]]
    final = env.get_state()
    other = crl.HipPongVecEnv(n, seed=3, mode="wrapped", resized_dim=42, frame_stack=4)  # a fresh process would do the same
    other.reset()
    other.load_state_dict(sd)
    for a, w in zip(acts[20:], want):
        got = other.step_device(a)
        assert all(torch.equal(x, y) for x, y in zip(got, w))
    assert other.get_state().tobytes() == final.tobytes()
    with pytest.raises(ValueError):
        crl.HipPongVecEnv(8, seed=3, mode="wrapped", resized_dim=42, frame_stack=4).load_state_dict(sd)
    env.close(), other.close()


@pytest.mark.parametrize("solver", ["box2d", "fma"])
def test_car_resume_from_state_dict_reproduces_the_run(solver):
    _need_gpu()
    import competitive_rl_amd as crl

    n = 64
    env = crl.HipCarVecEnv(n, seed=5, solver=solver)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(2)
    acts = [torch.rand((n, 2, 2), generator=g, device="cuda") * 2 - 1 for _ in range(50)]
    for a in acts[:20]:
        env.step_device(a)
    sd = _roundtrip(env.state_dict())
    assert sd["kind"] == "cCarRacing" and sd["solver"] == solver
    want = [tuple(t.clone() for t in env.step_device(a)) for a in acts[20:]]
    final = env.get_state()
    assert (final["episode"] == sd["env_state"]["episode"]).all(), "no env was reset in between: the tracks are the checkpoint's"
    env.load_state_dict(sd)  # the same context: its tracks are the ones the checkpoint was taken on
    for a, w in zip(acts[20:], want):
        got = env.step_device(a)
        assert all(torch.equal(x, y) for x, y in zip(got, w))
    assert env.get_state().tobytes() == final.tobytes()
    #  a checkpoint of the OTHER arithmetic of the island solver is refused, not silently resumed on another trajectory
    other = crl.HipCarVecEnv(n, seed=5, solver="fma" if solver == "box2d" else "box2d")
    other.reset()
    with pytest.raises(ValueError, match="solver"):
        other.load_state_dict(sd)
    other.load_state_dict(sd["env_state"])  # (a bare state array carries no claim about the solver: accepted)
    other.close()
    env.close()
