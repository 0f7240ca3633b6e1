"""Shared by the CPU and GPU tests of tests/golden/car_step_books.npz (CarRacing.step bookkeeping recorded from the
reference): turns one recorded row into an oracle env whose bookkeeping is the row's "pre" state."""
import ctypes as C
import os

import numpy as np

from oracle import car_oracle as co

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load():
    return np.load(os.path.join(G, "car_step_books.npz"))


_tracks = {}


def base_env(g, scenario):
    """A fresh oracle env on the scenario's track (rebuilt from the recorded draws by the pinned generator)."""
    if scenario not in _tracks:
        e = co.CarEnv()
        u = g[f"track_u/{scenario}"]
        assert e.reset(u, 0) == len(u) // 24
        _tracks[scenario] = e.buf.copy()
    e = co.CarEnv()
    e.buf[:] = _tracks[scenario]
    e.e = e.buf[0]
    return e


def place(e, car, x, y):
    """Car `car` at rest with its hull ORIGIN at (x, y), angle 0 (Car.__init__ placement rule)."""
    off = co.ENV_DT.fields["car"][1] + car * co.CAR_DT.itemsize
    co.lib().car_oracle_place.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int]
    co.lib().car_oracle_place(C.c_void_p(e.buf.ctypes.data + off), 0.0, float(x), float(y), 0)


def env_for_row(g, r):
    """Oracle env teacher-forced to row r's pre-step bookkeeping.  The scripted stand-in engine of the generator never
    moved anything but the hull origin, so the cars are placed at rest there; a one-car scenario parks car 1, done."""
    scenario = str(g["scenario"][r])
    e = base_env(g, scenario)
    assert int(e.e["trk"]["n"]) == int(g["ntiles"][r]), scenario
    players = int(g["players"][r])
    for c in range(2):
        if c < players:
            place(e, c, *g["pre_pos"][r][c])
            e.e["reward"][c], e.e["prev_reward"][c] = g["pre_reward"][r][c], g["pre_prev_reward"][r][c]
            e.e["tile_visited_count"][c], e.e["last_block"][c] = g["pre_visited_count"][r][c], g["pre_last_block"][r][c]
            e.e["done"][c] = g["pre_done"][r][c]
            e.e["visited"][c] = g["pre_visited"][r][c]
            e.e["wheel_tiles"][c] = g["pre_wheel_tiles"][r][c]
        else:
            place(e, c, 300.0, -300.0)
            e.e["done"][c] = 1
    e.e["step_count"] = g["pre_step_count"][r]
    e.e["inv_dt0"] = 50.0
    e.e["contacts_enabled"] = 0
    return e


def usable(g, r):
    """Rows the physics-bearing implementations can replay: with action_repeat > 1 the first step of a scenario starts
    ON the track (the scripted engine raised no contact there; a real one does, between two repeats)."""
    return not (int(g["repeat"][r]) > 1 and int(g["t"][r]) == 0)
