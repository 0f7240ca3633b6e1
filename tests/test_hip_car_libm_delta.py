"""The stated CarRacing deviation, measured ON THE DEVICE.

north_star asks for 1e-5 on CarRacing float state against the reference's CPU step.  What the reference's step would compute here is
the oracle's libm build (``liboracle_libm.so``: glibc sinf / cosf / sin / cos / atan2, as box2d-py and CPython call them); the HIP
kernels evaluate include/crl_rot.h / crl_f64.h instead and equal ``liboracle.so`` / ``liboracle_fma.so`` at tolerance 0.
``tests/test_oracle_libm_delta.py`` measures the distance between those CPU builds; this file measures HIP itself against the libm
build -- one ``world.Step`` from IDENTICAL states, teacher-forced by the libm oracle, both arithmetics of the island solver -- and
asserts the same per-field bounds: positions, angles and (default arithmetic) linear velocities inside 1e-5; a wheel's spin is the
stated deviation (one unit in the last place of a float32 sine times the wheel's inverse inertia, 134)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

BODY = ("cx", "cy", "a", "vx", "vy", "w")
SOLVERS = pytest.mark.parametrize("solver", ["box2d", "fma"])
VEL_BOUND = {"box2d": 1e-5, "fma": 6e-5}   # (tests/test_oracle_libm_delta.py VEL_BOUND)


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def _check(worst, solver, spin_bound):
    """north_star's 1e-5 holds for every body's position and angle and for the HULLS' linear velocities (default arithmetic).  Measured on
    these trajectories (HIP == the oracle's default / fused build bit for bit, so the numbers are those builds'): positions <= 1.2e-7, angles
    <= 1.3e-6 (fma 6.6e-6), hull velocity <= 8.4e-6 (fma 1.4e-5); a WHEEL's linear velocity reaches 4.5e-5 in single steps (fma 2.4e-5) and
    the angular velocities 1.4e-5 (fma 8.1e-5): one unit in the last place of a float32 sine behind a stiff joint and a wheel's inverse
    inertia (134) -- the stated deviation, bounded here, not hidden."""
    for part in ("hull", "wheel"):
        for f in ("cx", "cy", "a"):
            assert worst[f"{part}.{f}"] <= 1e-5, (part, f, worst)
    for f in ("vx", "vy"):
        assert worst[f"hull.{f}"] <= VEL_BOUND[solver], (f, worst)
        assert worst[f"wheel.{f}"] <= 1e-4, (f, worst)
    assert worst["hull.w"] <= spin_bound and worst["wheel.w"] <= spin_bound, worst


def _teacher_forced(solver, twins, actions_of, steps):
    """HIP env stepped from the libm oracle's state, every step; returns (max relative |d| per field, env-steps with contact)"""
    import competitive_rl_amd as crl
    from tests.test_hip_car_parity import oracle_to_hip_state, push_tracks

    n = len(twins)
    env = crl.HipCarVecEnv(n, solver=solver)
    env.reset()
    push_tracks(env, twins)
    worst, touched = {}, 0
    for t in range(steps):
        acts = actions_of(t)
        env.set_state(oracle_to_hip_state(twins))          # identical pre-step state
        _, rew, done = env.step_device(torch.as_tensor(acts).cuda(), render=False)
        for i, tw in enumerate(twins):
            tw.step(acts[i].astype(np.float64))
        st = env.get_state()
        for i, tw in enumerate(twins):
            for c in range(2):
                q, o = st[i]["car"][c], tw.e["car"][c]
                for f in BODY:
                    for part in ("hull", "wheel"):
                        worst[part + "." + f] = max(worst.get(part + "." + f, 0.0), _rel(q[part][f], o[part][f]))
                worst["imp"] = max(worst.get("imp", 0.0), _rel(q["imp"], o["imp"]))
                worst["omega64"] = max(worst.get("omega64", 0.0), _rel(q["omega"], o["omega"]))
                assert int(q["tile_visited_count"]) == int(tw.e["tile_visited_count"][c]) and int(q["done"]) == int(tw.e["done"][c]), (t, i, c)
            assert int(st[i]["n_contact"]) == int(tw.e["n_contact"]), (t, i)
            touched += int(tw.e["n_contact"]) > 0
    env.close()
    return worst, touched


@SOLVERS
def test_hip_free_driving_one_step_distance_to_the_libm_oracle(solver):
    _need_gpu()
    from tests.car_scenarios import make_oracle_envs

    n, steps = 12, 160
    twins = make_oracle_envs(n, libm=True)
    rs = np.random.RandomState(4)

    def acts(t):
        a = rs.uniform(-1, 1, (n, 2, 2)).astype(np.float32)
        if t < 60:
            a[:, :, 1] = np.abs(a[:, :, 1])
        return a

    worst, _ = _teacher_forced(solver, twins, acts, steps)
    print(f"HIP ({solver}) vs libm oracle, one step from identical state, free driving: max relative |d|", worst)
    _check(worst, solver, spin_bound=2e-3)
    assert worst["imp"] <= 2e-3, worst


@SOLVERS
def test_hip_touching_cars_one_step_distance_to_the_libm_oracle(solver):
    _need_gpu()
    from tests.car_scenarios import crash_actions, make_oracle_envs, park_for_crash

    n, steps = 8, 150
    twins = make_oracle_envs(n, seed0=20, libm=True)
    park_for_crash(twins)
    worst, touched = _teacher_forced(solver, twins, lambda t: crash_actions(n, t), steps)
    print(f"HIP ({solver}) vs libm oracle, one step from identical state, cars touching in {touched} env-steps: max relative |d|", worst)
    assert touched > 50
    _check(worst, solver, spin_bound=5e-3)
