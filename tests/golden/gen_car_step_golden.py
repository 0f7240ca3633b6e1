"""Golden vectors for ``CarRacing.step``'s own bookkeeping.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_car_step_golden.py

Drives the reference's ``CarRacing.reset`` / ``CarRacing.step``
(car_racing/car_racing_multi_players.py:454-620) with the Box2D stand-in of ``_car_standins.py``:
the engine does no physics, and what it "did" in each ``world.Step`` is SCRIPTED here -- where the
hulls are afterwards, and which wheel/tile Begin/EndContact events the reference's own
``FrictionDetector`` receives.  Everything that is recorded is computed by the reference's code:
the -0.1/action_repeat time penalty, the step-reward delta taken BEFORE the world step (tile rewards
are reported one step late), the three done rules and their order relative to ``step_count``,
done cars being skipped, ``info["num_steps"]``, action repetition.

``car_step_books.npz``: one row per recorded ``step`` call = (bookkeeping state before the call,
actions, script of the call) -> (step rewards, done flags, num_steps, bookkeeping state after).
Rendering is replaced by a stub (the observation is not part of this fixture).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _car_standins as B  # noqa: E402

MAXT = 512


def main():
    cd, cr = B.load_car_reference()

    class Env(cr.CarRacing):  # the reference's step/reset; only the drawing is stubbed
        def get_observation(self, i):
            return np.zeros((96, 96, 1), np.uint8)

        def camera_update(self, *a, **k):
            return None

        def render_road_for_observation_map(self, surface):
            return surface

    rows = []

    def snapshot(env):
        P = env.num_player
        d = dict(reward=np.zeros(2), prev_reward=np.zeros(2), visited_count=np.zeros(2, np.int32), done=np.zeros(2, np.int32),
                 last_block=np.full(2, -1, np.int32), pos=np.zeros((2, 2)), visited=np.zeros((2, MAXT // 32), np.uint32),
                 wheel_tiles=np.zeros((2, 4, MAXT // 32), np.uint32))
        for c in range(P):
            d["reward"][c], d["prev_reward"][c] = env.rewards[c], env.prev_rewards[c]
            d["visited_count"][c], d["done"][c] = env.tile_visited_count[c], int(env.done[c])
            if env.block_visited[c]:
                d["last_block"][c] = env.block_visited[c][-1]
            d["pos"][c] = env.cars[c].hull.position
            for t in env.road:
                if t.road_visited[c]:
                    d["visited"][c, t.block_id >> 5] |= np.uint32(1 << (t.block_id & 31))
            for w, wheel in enumerate(env.cars[c].wheels):
                for t in wheel.tiles:
                    d["wheel_tiles"][c, w, t.block_id >> 5] |= np.uint32(1 << (t.block_id & 31))
        d["step_count"] = env.step_count
        return d

    def run(name, players, repeat, seed, steps, script):
        """script(t, sub, env) -> (positions {car: (x, y)} or None, events [(car, wheel, tile, begin)]) applied in world.Step"""
        draws = []

        def stream(_seed):
            d = B.Draws(seed)
            draws.append(d)
            return d

        cr.seeding.np_random = lambda s=None: (stream(s), s)
        env = Env(num_player=players, verbose=0, action_repeat=repeat)
        draws_before = len(draws[-1].u)
        # reset() shuffles the birth places with the GLOBAL numpy generator (car_racing_multi_players.py:508-509): seeded here, or the recorded
        # start positions (+- 5 in x) differ from run to run (found by tests/test_golden_regenerates.py, round 6)
        np.random.seed(seed)
        env.reset()
        u = np.array(draws[-1].u[draws_before:])
        assert len(u) % 24 == 0
        ntiles = len(env.track)
        tiles = {t.block_id: t for t in env.road}
        rs = np.random.RandomState(seed)
        cur = {"t": 0, "sub": 0}

        def on_step(world):
            pos, events = script(cur["t"], cur["sub"], env)
            for (c, w, tid, begin) in events:
                wheel, tile = env.cars[c].wheels[w], tiles[tid]
                if begin == (tile in wheel.tiles):
                    continue  # Box2D raises Begin only for a new overlap and End only for an existing one
                fa = type("F", (), {})()
                fb = type("F", (), {})()
                fa.body, fb.body = type("Bd", (), {"userData": tile})(), type("Bd", (), {"userData": wheel})()
                contact = type("C", (), {"fixtureA": fa, "fixtureB": fb})()
                (world.listener.BeginContact if begin else world.listener.EndContact)(contact)
            if pos:
                for c, p in pos.items():
                    env.cars[c].hull.position = B.Vec2(p)
            cur["sub"] += 1

        B.World.on_step = on_step
        for t in range(steps):
            cur["t"], cur["sub"] = t, 0
            pre = snapshot(env)
            a = rs.uniform(-1.3, 1.3, (2, 2))
            if players == 2:
                o, r, d, info = env.step({0: a[0], 1: a[1]})
                rew = np.array([r[0], r[1]])
                done = np.array([d[0], d[1]], np.int32)
                ns = info[0]["num_steps"]
                assert info[1]["num_steps"] == ns and set(info) == {0, 1} and set(info[0]) == {"num_steps"}
            else:
                o, r, d, info = env.step(a[0])
                rew, done, ns = np.array([r, 0.0]), np.array([d, 0], np.int32), info["num_steps"]
                assert set(info) == {"num_steps"}
            post = snapshot(env)
            rows.append(dict(scenario=name, players=players, repeat=repeat, ntiles=ntiles, t=t, action=a, rew=rew, done_out=done, num_steps=ns,
                             track_u=u, pre=pre, post=post))
        B.World.on_step = None
        return env

    # ---- S1: a lap.  Car 0 takes one new tile per step (wheels in turn, leaving the previous one), car 1 every
    # third step; car 0 finishes the lap and is skipped from then on while car 1 keeps going.
    def lap(t, sub, env):
        n = len(env.track)
        ev = []
        if t < n:
            ev.append((0, t % 4, t, True))
            if t >= 4:
                ev.append((0, t % 4, t - 4, False))
        if t % 3 == 0 and t // 3 < n:
            ev.append((1, (t // 3) % 4, t // 3, True))
        if t % 7 == 0:  # a hull/tile contact is ignored; a second Begin on a visited tile pays nothing
            ev.append((1, 0, 0, True))
            ev.append((1, 0, 0, False))
        return None, ev

    run("lap", 2, 1, 101, 330, lap)

    # ---- S2: leaving the playfield (|x| or |y| > 2000/6), one car after the other
    def out(t, sub, env):
        pos = {0: (100.0 + 2.0 * t, -50.0), 1: (-20.0, 300.0 + 1.7 * t)}
        ev = [(0, 1, t // 2, True)] if t % 2 == 0 and t // 2 < 40 else []
        return pos, ev

    run("out", 2, 1, 102, 130, out)

    # ---- S3: the step_count > 1000 rule (nothing else happens)
    run("timeout", 2, 1, 103, 1004, lambda t, sub, env: (None, []))

    # ---- S4: action_repeat = 4: the penalty is 0.1/4 per repeat, step_count crosses 1000 inside a step
    def idle_far(t, sub, env):
        return ({0: (300.0, 300.0), 1: (-300.0, 300.0)}, []) if sub == 3 else (None, [])

    run("repeat4_timeout", 2, 4, 104, 254, idle_far)

    # ---- S5: action_repeat = 2: a car leaves the field between two steps; tile events in the last repeat only
    def out2(t, sub, env):
        if sub != 1:
            return None, []
        pos = {0: (300.0, 250.0 + 3.0 * t), 1: (-300.0 - 1.5 * t, 300.0)}
        ev = [(1, 2, t, True)] if t < 30 else []
        return pos, ev

    run("repeat2_out", 2, 2, 105, 60, out2)

    # ---- S6: cCarRacing-v0 (one car): scalar returns
    def lap1(t, sub, env):
        n = len(env.track)
        ev = [(0, t % 4, min(2 * t, n - 1), True), (0, (t + 1) % 4, min(2 * t + 1, n - 1), True)] if 2 * t < n else []
        pos = {0: (50.0, -340.0)} if t == 200 else None
        return pos, ev

    run("single", 1, 1, 106, 215, lap1)

    out_ = {"count": len(rows), "scenario": np.array([r["scenario"] for r in rows])}
    for k in ("players", "repeat", "ntiles", "t", "num_steps"):
        out_[k] = np.array([r[k] for r in rows], np.int32)
    for k in ("action", "rew", "done_out"):
        out_[k] = np.array([r[k] for r in rows])
    for side in ("pre", "post"):
        for k in rows[0][side]:
            out_[f"{side}_{k}"] = np.array([r[side][k] for r in rows])
    # one track per scenario: its draws (the tests rebuild it with the pinned track generator)
    names = sorted(set(out_["scenario"].tolist()))
    for nm in names:
        out_[f"track_u/{nm}"] = next(r["track_u"] for r in rows if r["scenario"] == nm)
    np.savez_compressed(os.path.join(HERE, "car_step_books.npz"), **out_)
    for nm in names:
        m = out_["scenario"] == nm
        print(f"{nm}: {int(m.sum())} steps, ntiles {int(out_['ntiles'][m][0])}, done steps {int((out_['done_out'][m].sum(1) > 0).sum())}, "
              f"reward range [{out_['rew'][m].min():.3f}, {out_['rew'][m].max():.3f}], attempts {len(out_[f'track_u/{nm}']) // 24}")


if __name__ == "__main__":
    main()
