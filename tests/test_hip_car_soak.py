"""The pipelined CarRacing step at full size with NO host synchronisation inside the loop (a read-back per step would hide a cross-stream
race): the bench's steady-state workload -- staggered TimeLimit, 16 cycled action tensors, 7 % of the envs touching, 16 resets per step --
run three times: pipelined twice (determinism) and on one stream with everything in place (CRL_CAR_NO_OVERLAP=1).  Frame checksums, rewards
and done counts of every step are accumulated on the device and must agree step for step; so must the complete state at the end.
(tools/car_soak_nosync.py is the same for 3 000 steps: 49 228 episode ends, no difference, both arithmetics.)"""
import os

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("solver", ["box2d", "fma"])
def test_pipelined_steps_without_host_sync_equal_the_one_stream_step(solver):
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    import competitive_rl_amd as crl

    n, steps = 16384, 700

    def run():
        env = crl.HipCarVecEnv(n, seed=21, solver=solver)
        env.reset()
        st = env.get_state()
        st["elapsed"] = (torch.arange(n, dtype=torch.int64) * 1000 // n).numpy().astype(st["elapsed"].dtype)
        env.set_state(st)
        g = torch.Generator(device="cuda").manual_seed(4)
        acts = torch.rand((16, n, 2, 2), generator=g, device="cuda") * 2 - 1
        w = torch.arange(1, 96 * 96 + 1, device="cuda", dtype=torch.int64)
        sig = torch.zeros((steps, 3), dtype=torch.float64, device="cuda")
        for t in range(steps):
            obs, rew, done = env.step_device(acts[t % 16])
            sig[t, 0] = (obs.view(n, 2, -1).to(torch.int64) * w).sum().double()
            sig[t, 1] = rew.double().sum()
            sig[t, 2] = done.sum().double()
        torch.cuda.synchronize()
        st, caps = env.get_state(), env.cap_hits()
        env.close()
        return sig.cpu(), st, caps

    a, sa, ca = run()
    b, sb, _ = run()
    os.environ["CRL_CAR_NO_OVERLAP"] = "1"  # (read when the context is created)
    try:
        c, sc, _ = run()
    finally:
        del os.environ["CRL_CAR_NO_OVERLAP"]
    assert ca == (0, 0, 0, 0)
    assert int(a[:, 2].sum()) > 8000 and int((sa["n_contact"] > 0).sum()) > 500, "the workload must reset and touch"
    bad_ab, bad_ac = torch.nonzero((a != b).any(1)).reshape(-1), torch.nonzero((a != c).any(1)).reshape(-1)
    assert not len(bad_ab), ("two pipelined runs differ from step", int(bad_ab[0]))
    assert not len(bad_ac), ("pipelined and one-stream runs differ from step", int(bad_ac[0]))
    assert sa.tobytes() == sb.tobytes() == sc.tobytes()
