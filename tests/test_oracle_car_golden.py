"""CarRacing oracle against golden vectors recorded from the reference's own Python
(tests/golden/gen_car_golden.py): track generator, wheel model, action mapping, tile rule."""
import os

import numpy as np
import pytest

from oracle import car_oracle as co

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_track_generation_bit_exact():
    """liboracle_libm.so: same sources as liboracle.so with math.sin / cos / atan2 = the host libm's, as in CPython.
    (liboracle.so's shared evaluations differ from it by <= 1e-13 on these tracks: tests/test_oracle_libm_delta.py.)"""
    g = np.load(os.path.join(G, "car_track.npz"))
    n_ok = 0
    for j in range(int(g["count"])):
        ok, trk = co.create_track(g[f"{j}/draws"], libm=True)
        assert ok == bool(g[f"{j}/ok"]), j
        if not ok:
            continue
        n_ok += 1
        want = g[f"{j}/track"]
        n = int(trk["n"])
        assert n == len(want), j
        # glibc libm here is what CPython's math module calls: bit-exact f64
        assert np.array_equal(trk["track"][:n].view(np.uint64), want.view(np.uint64)), j
        assert np.array_equal(trk["border"][:n], g[f"{j}/border"]), j
        tiles = g[f"{j}/tiles"]
        if len(tiles):
            # reference creation order is i = n-1 .. 0
            assert np.array_equal(trk["tile"][:n][::-1].view(np.uint64), tiles.view(np.uint64)), j
            bp, b = g[f"{j}/border_poly"], g[f"{j}/border"].astype(bool)
            assert np.array_equal(trk["border_poly"][:n][b].view(np.uint64), bp[b].view(np.uint64)), j
    assert n_ok >= 40


def test_wheel_model_bit_exact():
    g = np.load(os.path.join(G, "car_wheels.npz"))
    inp, out = g["inp"], g["out"]
    car = np.zeros(1, co.CAR_DT)
    for r in range(len(inp)):
        steer, gas, brake = inp[r, :3]
        w_in = inp[r, 3:].reshape(4, 9)
        w_out = out[r].reshape(4, 8)
        car["gas"][0] = w_in[:, 8]
        co.lib().car_oracle_controls(car.ctypes.data, float(steer), float(gas), float(brake))
        for w in range(4):
            qs, qc, vx, vy, ja, omega, phase, on_road, _ = w_in[w]
            om, ph, ms, f = co.wheel(1.0 / 50, car["steer"][0][w], car["gas"][0][w], car["brake"][0][w], ja, qs, qc, vx, vy,
                                     on_road > 0, omega, phase)
            want = w_out[w]
            got = np.array([om, ph, ms, f[0], f[1], car["gas"][0][w], car["brake"][0][w], car["steer"][0][w]])
            assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (r, w, got, want)


def test_process_action_and_tile_rule():
    g = np.load(os.path.join(G, "car_rules.npz"))
    for a, want in zip(g["actions"], g["processed"]):
        assert np.array_equal(co.process_action(a), want)
    env = co.CarEnv()
    env.e["trk"]["n"] = int(g["track_len"])
    env.e["last_block"][:] = -1
    for (c, w, t, begin), want in zip(g["events"], g["results"]):
        if w >= 0:  # hull-tile contacts are ignored by the listener (no "tiles" attribute)
            env.contact_event(int(c), int(w), int(t), int(begin))
        got = [env.e["reward"][0], env.e["reward"][1], env.e["tile_visited_count"][0], env.e["tile_visited_count"][1]]
        assert got == want.tolist()


def test_body_constants_match_survey():
    k = co.consts()
    # SURVEY D.1 (computed independently with f64 polygon formulas)
    assert abs(k["hull_mass"] - 7.06) < 1e-3 and abs(k["hull_inv_I"] - 0.054908) < 1e-5
    assert abs(k["hull_lc"][1] + 0.08253) < 1e-4 and abs(k["hull_lc"][0]) < 1e-6
    assert abs(k["wheel_mass"] - 0.06048) < 1e-6 and abs(k["wheel_inv_I"] - 134.0626) < 1e-2
