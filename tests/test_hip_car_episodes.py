"""cCarRacingDouble: the PRODUCTION step pipeline (streams, staged resets, collide-ahead, frames drawn by the touching solve)
against the CPU oracle THROUGH EPISODE ENDS, and the long teacher-forced soak.

The other parity tests either stop comparing an env when its episode ends or compare the pipelined step with the one-stream
step of the same library.  Here oracle and HIP env go through `done -> terminal frame -> reset -> first frame`
(utils/dummy_vec_env.py:55-58, car_racing_multi_players.py:454-525) together, several episodes per env, some of them ending
while the cars touch: bodies, rewards, done flags, info["terminal_observation"], the new track, the map in its new slot and
the first frame of the new episode -- tolerance 0 (both sides evaluate include/crl_rot.h / crl_f64.h)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

BODY = ("cx", "cy", "a", "vx", "vy", "w")


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")


def batch_to_hip_state(E, elapsed=None, episode=None):
    """oracle car_env array (oracle.car_oracle.ENV_DT) -> crl_car_env_state array, vectorised over the envs"""
    from competitive_rl_amd import _native as N

    st = np.zeros(len(E), N.CAR_ENV_STATE_DT)
    for c in range(2):
        q, o = st["car"][:, c], E["car"][:, c]
        for f in BODY:
            q["hull"][f] = o["hull"][f]
            q["wheel"][f] = o["wheel"][f]
        for f in ("imp", "motor_imp", "motor_speed", "limit_state", "gas", "omega", "phase", "sleep_time"):
            q[f] = o[f]
        q["reward"], q["prev_reward"] = E["reward"][:, c], E["prev_reward"][:, c]
        q["tile_visited_count"], q["last_block"], q["done"] = E["tile_visited_count"][:, c], E["last_block"][:, c], E["done"][:, c]
        q["step_count"], q["first_step"] = E["step_count"], (E["inv_dt0"] == 0).astype(np.int32)
        q["wheel_tiles"], q["visited"] = E["wheel_tiles"][:, c], E["visited"][:, c]
    st["elapsed"] = E["step_count"] if elapsed is None else elapsed
    if episode is not None:
        st["episode"] = episode  # resets so far: indexes the replayed / Philox draws of the env's NEXT track
    st["n_contact"] = E["n_contact"]
    for f in ("pair", "count", "type", "ln", "lp", "pt", "id", "nimp", "timp"):
        st["contact"][f] = E["contact"][f]
    return st


def assert_state_equal(hs, E, where, contacts=True, only=None):
    """HIP state (get_state) against the oracle's, bit for bit: bodies, joints, wheel model, tile bookkeeping, manifolds"""
    sel = slice(None) if only is None else only
    for c in range(2):
        q, o = hs["car"][sel, c], E["car"][sel, c]
        for f in BODY:
            for part in ("hull", "wheel"):
                bad = q[part][f] != o[part][f]
                assert not bad.any(), (where, "car", c, part, f, "envs", np.nonzero(bad.reshape(len(bad), -1).any(1))[0][:8])
        for f in ("imp", "motor_imp", "limit_state", "gas", "omega", "phase", "sleep_time"):
            bad = (q[f] != o[f]).reshape(len(q), -1).any(1)
            assert not bad.any(), (where, "car", c, f, "envs", np.nonzero(bad)[0][:8])
        assert np.array_equal(q["tile_visited_count"], E["tile_visited_count"][sel, c]), (where, c, "tile_visited_count")
        assert np.array_equal(q["visited"], E["visited"][sel, c]) and np.array_equal(q["wheel_tiles"], E["wheel_tiles"][sel, c]), (where, c, "tiles")
        assert np.array_equal(q["last_block"], E["last_block"][sel, c]) and np.array_equal(q["done"], E["done"][sel, c]), (where, c)
        assert np.array_equal(q["reward"], E["reward"][sel, c]), (where, c, "reward")
    if contacts:
        assert np.array_equal(hs["n_contact"][sel], E["n_contact"][sel]), (where, "n_contact")
        for k in range(8):
            live = E["n_contact"][sel] > k
            q, o = hs["contact"][sel][:, k][live], E["contact"][sel][:, k][live]
            for f in ("pair", "count", "type"):
                assert np.array_equal(q[f], o[f]), (where, "contact", k, f)
            for j in range(2):
                pt = o["count"] > j
                for f in ("id", "nimp", "timp"):
                    assert np.array_equal(q[f][pt, j], o[f][pt, j]), (where, "contact", k, f, j)


def _park_beside(E, idx, gap=2.6):
    """car 1 of the listed envs beside car 0, wheels overlapping: the narrow phase finds manifolds at once"""
    c0, c1 = E["car"][:, 0], E["car"][:, 1]
    ang = c0["hull"]["a"][idx].astype(np.float64)
    dx, dy = (np.cos(ang) * gap).astype(np.float32), (np.sin(ang) * gap).astype(np.float32)
    for f in ("a", "vx", "vy", "w"):
        c1["hull"][f][idx] = c0["hull"][f][idx]
        c1["wheel"][f][idx] = c0["wheel"][f][idx]
    c1["hull"]["cx"][idx], c1["hull"]["cy"][idx] = c0["hull"]["cx"][idx] + dx, c0["hull"]["cy"][idx] + dy
    c1["wheel"]["cx"][idx], c1["wheel"]["cy"][idx] = c0["wheel"]["cx"][idx] + dx[:, None], c0["wheel"]["cy"][idx] + dy[:, None]


# the island solver's two arithmetics, each against its own oracle build (tests/test_hip_car_parity.py: SOLVERS)
SOLVERS = pytest.mark.parametrize("solver", ["box2d", "fma"])
ORACLE_OF = {"box2d": False, "fma": "fma"}


@SOLVERS
def test_pipelined_step_equals_the_oracle_through_episode_ends(solver):
    _need_gpu()
    import competitive_rl_amd as crl
    from competitive_rl_amd import _native as N
    from oracle import car_oracle as co

    co.set_text(N.load_car_text())
    n, phases, steps, A = 64, 3, 48, 16  # A attempts per episode (car_track.hip: attempt a of episode e = draws[(16 e + a) % attempts])
    rs = np.random.RandomState(77)
    u = rs.random_sample((n, (phases + 1) * A, 24))
    swap = rs.randint(0, 2, (n, (phases + 1) * A)).astype(np.uint8)
    hip = crl.HipCarVecEnv(n, seed=1, solver=solver)  # default context: the pipelined step
    hip.set_replay(u, swap)
    B = co.CarBatch(n, libm=ORACLE_OF[solver])
    episode = np.zeros(n, np.int64)

    def oracle_reset(i):
        o, ep = B.view(i), int(episode[i])
        draws = u[i, ep * A:(ep + 1) * A].reshape(-1)
        att = o.reset(draws, 0)
        assert att > 0, ("no attempt of the replayed draws closes a lap", i, ep)
        o.reset(draws, int(swap[i, ep * A + att - 1]))
        o.e["contacts_enabled"] = 1
        o.step(None)
        episode[i] += 1

    obs = hip.reset().cpu().numpy()
    for i in range(n):
        oracle_reset(i)
        for v in range(2):
            assert np.array_equal(obs[i, v], B.view(i).render(v)), ("first frame after reset()", i, v)
    elapsed = np.zeros(n, np.int64)
    seen = dict(finished=np.zeros(n, np.int64), finished_touching=0, finished_coupled=0, touching_steps=0, frames=0, tracks=0, maps=0)
    for ph in range(phases):
        # stagger the TimeLimit so that every env finishes inside this phase, at its own step; a third of the envs with the cars
        # wheel to wheel (they finish while their island is being solved), the state pushed from the oracle
        assert_state_equal(hip.get_state(), B.E, ("phase start", ph))
        touch = np.nonzero(np.arange(n) % 3 == ph % 3)[0]
        _park_beside(B.E, touch)
        elapsed[:] = 1000 - 4 - (np.arange(n) * 11 + 5 * ph) % (steps - 8)
        hip.set_state(batch_to_hip_state(B.E, elapsed, episode))
        for t in range(steps):
            acts = rs.uniform(-1, 1, (n, 2, 2)).astype(np.float32)
            acts[:, 1, 0] = -acts[:, 0, 0] * (t % 7 < 4)  # the parked cars keep bumping into each other
            obs, rew, done = hip.step_device(torch.as_tensor(acts).cuda())
            obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy().astype(bool)
            coupled_before = B.E["n_contact"] > 0
            r, d = B.step(acts.astype(np.float64))
            elapsed += 1
            want_done = d.any(1) | (elapsed >= 1000)
            assert np.array_equal(rew, r.astype(np.float32)), (ph, t, np.nonzero((rew != r.astype(np.float32)).any(1))[0][:8])
            assert np.array_equal(done, want_done), (ph, t, np.nonzero(done != want_done)[0])
            seen["touching_steps"] += int((B.E["n_contact"] > 0).sum())
            fin = np.nonzero(want_done)[0]
            if len(fin):
                term = hip.terminal_observation(torch.as_tensor(fin).cuda())
                for k, i in enumerate(fin):
                    tob = term[k].cpu().numpy()
                    for v in range(2):
                        assert np.array_equal(tob[v], B.view(i).render(v)), ("terminal_observation", ph, t, i, v)
                    seen["finished_touching"] += int(B.E["n_contact"][i] > 0)
                    seen["finished_coupled"] += int(coupled_before[i])
                    oracle_reset(i)  # DummyVecEnv: obs = env.reset() (dummy_vec_env.py:55-58)
                    elapsed[i] = 0
                    seen["finished"][i] += 1
                    tr, nt = hip.get_track(int(i)), int(B.E[i]["trk"]["n"])
                    assert tr["n"] == nt and np.array_equal(tr["tile_poly"], B.E[i]["tile32"][:nt]), ("new track", ph, t, i)
                    seen["tracks"] += 1
                    if seen["maps"] < 12 or B.E["n_contact"][i] > 0:  # (1.5 M pixels per map: a sample, and every env that finished touching)
                        m, overflow = hip.get_map(int(i))
                        assert overflow == 0 and np.array_equal(m, B.view(i).map()), ("map of the new episode", ph, t, i)
                        seen["maps"] += 1
            for i in range(n):  # every frame of every step: the first frame of a new episode for the envs that finished
                for v in range(2):
                    want = B.view(i).render(v)
                    assert np.array_equal(obs[i, v], want), ("frame", ph, t, i, v, bool(want_done[i]), int((obs[i, v] != want).sum()))
                    seen["frames"] += 1
            if t % 6 == 5 or len(fin):
                assert_state_equal(hip.get_state(), B.E, (ph, t))
        assert (seen["finished"] >= ph + 1).all(), ("every env finishes once per phase", ph, np.nonzero(seen["finished"] < ph + 1)[0])
    assert hip.cap_hits() == (0, 0, 0, 0)
    print("episodes through the pipeline:", {k: (int(v.sum()) if hasattr(v, "sum") else v) for k, v in seen.items()})
    assert seen["finished_touching"] >= 3 and seen["touching_steps"] > 200, seen
    co.set_text(None)
    hip.close()


@SOLVERS
def test_teacher_forced_soak_2048_envs_200_steps(solver):
    """tools/car_teacher_soak.py as a test, at scale: every step starts from the oracle's state (pushed as a whole), both
    sides step once -- the oracle over the host's cores (car_oracle_step_batch) --, and the complete state is compared with
    tolerance 0: bodies, joint impulses, sleep timers, wheel model, tile bookkeeping, manifolds and their impulses.  An env whose
    episode ends is put back to its first state on both sides (the soak is about the step, not about resets)."""
    _need_gpu()
    import competitive_rl_amd as crl
    from oracle import car_oracle as co
    from tests.car_scenarios import make_oracle_envs

    n, steps, base = 2048, 200, 32
    seeds = make_oracle_envs(base, seed0=5, libm=ORACLE_OF[solver])  # 32 distinct tracks; env i starts from seed i % 32, then diverges (own actions)
    B = co.CarBatch(n, libm=ORACLE_OF[solver])
    for i in range(n):
        B.E[i] = seeds[i % base].e
    start = B.E.copy()
    hip = crl.HipCarVecEnv(n, solver=solver)
    hip.reset()

    def push_track(j):
        e = seeds[j % base].e
        nt = int(e["trk"]["n"])
        border = np.where(e["trk"]["border"][:nt] > 0, np.where(np.arange(nt) % 2 == 0, 1, 2), 0).astype(np.uint8)
        hip.set_track(j, e["trk"]["tile"][:nt], e["trk"]["border_poly"][:nt], border, e["trk"]["track"][0][1:4].astype(np.float32))

    for j in range(n):
        push_track(j)
    rs = np.random.RandomState(12)
    # a quarter of the envs start with the cars wheel to wheel
    _park_beside(B.E, np.nonzero(np.arange(n) % 4 == 1)[0])
    touching = restarts = 0
    for t in range(steps):
        hip.set_state(batch_to_hip_state(B.E))
        acts = rs.uniform(-1, 1, (n, 2, 2)).astype(np.float32)
        if (t // 50) % 2 == 0:
            acts[:, :, 1] = np.abs(acts[:, :, 1])
        if t % 97 < 30:
            acts[:, 1] = acts[:, 0]  # both cars do the same: they stay close and bump into each other
        _, rew, _ = hip.step_device(torch.as_tensor(acts).cuda(), render=False)
        r, d = B.step(acts.astype(np.float64))
        assert np.array_equal(rew.cpu().numpy(), r.astype(np.float32)), t
        assert_state_equal(hip.get_state(), B.E, ("soak", t))
        touching += int((B.E["n_contact"] > 0).sum())
        over = np.nonzero(d.any(1))[0]
        B.E[over] = start[over]
        for j in over:  # (the HIP env has auto-reset onto a new track: back to the recorded one)
            push_track(int(j))
        restarts += len(over)
    print("teacher-forced soak:", n, "envs x", steps, "steps; touching env-steps", touching, "restarts", restarts)
    assert touching > 20000 and hip.cap_hits() == (0, 0, 0, 0)
    hip.close()
