"""Golden vectors for the CarRacing pieces of the hot path that live in the REFERENCE's own
Python (not in Box2D): track generation, the per-wheel engine/brake/friction model, the
action mapping and the tile-visit reward rule.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_car_golden.py

Loads ``car_racing/car_racing_multi_players.py`` and ``car_racing/car_dynamics.py`` by path
with stand-ins for Box2D / pygame / gym / matplotlib (none installable here).  Recorded:

* ``car_track.npz``   -- ``CarRacing._create_track`` (:262-452): the uniform draws of every
  attempt -> success flag, track (alpha, beta, x, y) as f64, border flags, tile polygons.
* ``car_wheels.npz``  -- ``Car.gas/brake/steer`` + ``Car.step`` (car_dynamics.py:131-234) on
  stand-in bodies: inputs (controls, wheel state, body velocity/angle, on-road flag) ->
  omega, phase, joint.motorSpeed, applied force.
* ``car_rules.npz``   -- ``CarRacing.process_action`` (:527-540) samples and
  ``FrictionDetector._contact`` (:111-153) event sequences -> rewards / visit counts.

The physics engine itself (b2World.Step) is third-party and cannot be recorded: parity for
it is UNPINNED.
"""
import math
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_standins as S  # noqa: E402


from _car_standins import Draws, FixtureDef, Shape, Vec2, World, load_car_reference  # noqa: E402


class Obj:
    """hashable attribute bag (tiles / wheels / hulls in the rule tests)"""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def main():
    cd, cr = load_car_reference()

    # ---------------------------------------------------------------- tracks
    env = cr.CarRacing.__new__(cr.CarRacing)
    env.verbose, env.num_player, env.road, env.world = 0, 2, None, World()
    T_MAX = 400
    recs = []
    for seed in range(60):
        d = Draws(9000 + seed)
        env.np_random = d
        env.road_poly, env.road = [], []
        env.world = World()
        env.fd_tile = FixtureDef(shape=Shape(vertices=[(0, 0), (1, 0), (1, -1), (0, -1)]))
        ok = env._create_track()
        rec = dict(draws=np.array(d.u), ok=bool(ok))
        if ok:
            tr = np.array(env.track, np.float64)
            tiles = np.array([[list(v) for v in b.fixtures[0].shape.vertices] for b in env.road], np.float64)
            # border flags in tile creation order (i = len-1 .. 0): a 4-vertex road_poly entry follows its tile
            border = np.zeros(len(tr), np.uint8)
            side_col = np.zeros(len(tr), np.uint8)
            bpoly = np.zeros((len(tr), 4, 2), np.float64)
            i, k = len(tr) - 1, 0
            while k < len(env.road_poly):
                poly, color = env.road_poly[k]
                assert len(poly) == 5
                if k + 1 < len(env.road_poly) and len(env.road_poly[k + 1][0]) == 4:
                    border[i] = 1
                    bpoly[i] = np.array(env.road_poly[k + 1][0])
                    side_col[i] = 1 if tuple(env.road_poly[k + 1][1]) == (1, 1, 1) else 2
                    k += 1
                k += 1
                i -= 1
            assert i == -1
            if seed >= 8:  # full polygons only for the first tracks (fixture size)
                tiles, bpoly = tiles[:0], bpoly[:0]
            rec.update(track=tr, tiles=tiles, border=border, border_poly=bpoly, border_color=side_col)
        recs.append(rec)
    n_ok = sum(r["ok"] for r in recs)
    out = {"count": len(recs)}
    for j, r in enumerate(recs):
        for k, v in r.items():
            out[f"{j}/{k}"] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, "car_track.npz"), **out)
    print("tracks:", len(recs), "ok:", n_ok, "lens:", sorted(len(r["track"]) for r in recs if r["ok"])[::10])

    # ---------------------------------------------------------------- wheel model
    rs = np.random.RandomState(77)
    world = World()
    rows_in, rows_out = [], []
    for trial in range(400):
        car = cd.Car(world, float(rs.uniform(-3, 3)), 10.0, 20.0, 0, 0)
        for step in range(6):
            steer, gas, brake = float(rs.uniform(-1, 1)), float(rs.uniform(0, 1)), float(rs.choice([0, 0, 0.3, 0.95, rs.uniform(0, 1)]))
            if rs.random_sample() < 0.3:
                gas = 0.0
            pre = []
            for w in car.wheels:
                w.angle = float(np.float32(rs.uniform(-3.2, 3.2)))
                v = rs.uniform(-40, 40, 2) * (rs.random_sample() < 0.9)
                w.linearVelocity = Vec2(float(np.float32(v[0])), float(np.float32(v[1])))
                w.joint.angle = float(np.float32(rs.uniform(-0.45, 0.45)))
                w.omega = float(rs.uniform(-80, 120)) if rs.random_sample() < 0.9 else 0.0
                w.phase = float(rs.uniform(0, 50))
                on_road = rs.random_sample() < 0.6
                w.tiles = {Obj(road_friction=1.0)} if on_road else set()
                w.forces = []
                pre.append([float(np.float32(math.sin(w.angle))), float(np.float32(math.cos(w.angle))), w.linearVelocity[0], w.linearVelocity[1], w.joint.angle, w.omega, w.phase, float(on_road), w.gas])
            car.steer(steer), car.gas(gas), car.brake(brake)
            car.step(1.0 / 50)
            post = []
            for w in car.wheels:
                fx, fy = w.forces[-1]
                post.append([w.omega, w.phase, w.joint.motorSpeed, fx, fy, w.gas, w.brake, w.steer])
            rows_in.append([steer, gas, brake] + sum(pre, []))
            rows_out.append(sum(post, []))
    np.savez_compressed(os.path.join(HERE, "car_wheels.npz"), inp=np.array(rows_in, np.float64), out=np.array(rows_out, np.float64),
                        in_fields=np.array(["steer", "gas", "brake"] + [f"w{i}_{n}" for i in range(4) for n in
                                                                         ("q_sin", "q_cos", "vx", "vy", "joint_angle", "omega", "phase", "on_road", "gas_prev")]),
                        out_fields=np.array([f"w{i}_{n}" for i in range(4) for n in ("omega", "phase", "motor_speed", "fx", "fy", "gas", "brake", "steer")]))
    print("wheel cases:", len(rows_in))

    # ---------------------------------------------------------------- rules
    acts = rs.uniform(-1.6, 1.6, (200, 2))
    acts[:10] = [[0, 0], [1, 1], [-1, -1], [0.5, -0.5], [2, 2], [-2, -2], [0, 1e-9], [0, -1e-9], [1, 0], [-1, 0]]
    pa = np.array([cr.CarRacing.process_action(a) for a in acts], np.float64)

    # FrictionDetector._contact: synthetic begin/end sequences for 2 cars on a 120-tile track
    env2 = types.SimpleNamespace(ontrack_count=0, track=list(range(120)), rewards={0: 0.0, 1: 0.0},
                                 block_visited=[[], []], tile_visited_count={0: 0, 1: 0})
    fd = cr.FrictionDetector(env2)
    tiles = []
    for i in range(120):
        t = Obj(road_friction=1.0, block_id=i, color=[0, 0, 0], road_visited=[False, False])
        tiles.append(t)
    wheels = [[Obj(car_number=c, tiles=set()) for _ in range(4)] for c in range(2)]
    hulls = [Obj(car_number=c) for c in range(2)]
    ev, res = [], []
    pos = [0, 0]
    for t in range(1500):
        c = int(rs.randint(2))
        kind = rs.random_sample()
        if kind < 0.08:  # hull touches a tile: ignored (no "tiles" attribute)
            obj, tile, begin, wi = hulls[c], tiles[pos[c] % 120], True, -1
        else:
            wi = int(rs.randint(4))
            obj = wheels[c][wi]
            if obj.tiles and rs.random_sample() < 0.45:
                tile, begin = sorted(obj.tiles, key=lambda q: q.block_id)[0], False
            else:
                pos[c] += int(rs.choice([0, 0, 1, 1, 1, 2, 3]))
                tile, begin = tiles[min(pos[c], 119)], True
                if tile in obj.tiles:
                    continue
        fix = lambda o: types.SimpleNamespace(body=types.SimpleNamespace(userData=o))  # noqa: E731
        a, b = (fix(tile), fix(obj)) if rs.random_sample() < 0.5 else (fix(obj), fix(tile))
        fd._contact(types.SimpleNamespace(fixtureA=a, fixtureB=b), begin)
        ev.append([c, wi, tile.block_id, int(begin)])
        res.append([env2.rewards[0], env2.rewards[1], env2.tile_visited_count[0], env2.tile_visited_count[1]])
    np.savez_compressed(os.path.join(HERE, "car_rules.npz"), actions=acts, processed=pa, events=np.array(ev, np.int32),
                        results=np.array(res, np.float64), track_len=120)
    print("rule events:", len(ev), "visited:", env2.tile_visited_count)


if __name__ == "__main__":
    main()
