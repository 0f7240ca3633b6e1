"""Oracle-side scenario builders shared by the CPU and GPU CarRacing tests."""
import os

import numpy as np

from oracle import car_oracle as co

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def make_oracle_envs(n, seed0=0, libm=False):
    """n oracle envs on tracks rebuilt from the recorded draws of car_track.npz, after reset()'s step(None)."""
    g = np.load(os.path.join(G, "car_track.npz"))
    draws = [g[f"{j}/draws"] for j in range(int(g["count"]))]
    envs = []
    for i in range(n):
        u = np.concatenate([draws[(seed0 + i * 3 + k) % len(draws)] for k in range(6)])
        e = co.CarEnv(libm=libm)
        assert e.reset(u, i % 2) > 0
        e.e["contacts_enabled"] = 1
        e.step(None)
        envs.append(e)
    return envs


def park_for_crash(envs, seed=1):
    """Car 1 a few units ahead of car 0, slightly off-axis, joints settled: driving car 0 forward makes them touch."""
    rs = np.random.RandomState(seed)
    for i, e in enumerate(envs):
        c0, c1 = e.e["car"][0], e.e["car"][1]
        a = float(c0["hull"]["a"])
        hd, lat = np.array([-np.sin(a), np.cos(a)]), np.array([np.cos(a), np.sin(a)])
        tgt = np.array([c0["hull"]["cx"], c0["hull"]["cy"]]) + (7.0 + 0.3 * i) * hd + rs.uniform(-1.2, 1.2) * lat
        off = tgt - np.array([c1["hull"]["cx"], c1["hull"]["cy"]])
        c1["hull"]["cx"] += off[0]
        c1["hull"]["cy"] += off[1]
        for w in range(4):
            c1["wheel"][w]["cx"] += off[0]
            c1["wheel"][w]["cy"] += off[1]
        for k in range(30):  # let the joints settle before the crash
            e.step([[0.0, 0.0], [0.0, 0.0]])


def crash_actions(n, t):
    acts = np.zeros((n, 2, 2), np.float32)
    acts[:, 0, 1] = 1.0
    acts[:, 0, 0] = 0.2 * np.sin(t / 11.0)
    acts[:, 1, 1] = -0.3 if t > 90 else 0.0
    return acts
