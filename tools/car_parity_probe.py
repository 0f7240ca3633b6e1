#!/usr/bin/env python3
"""Teacher-forced HIP-vs-oracle CarRacing run that only REPORTS: worst relative state / impulse differences with and
without car-car contacts, and how many values are bit-identical (run on the GPU box)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import competitive_rl_amd as crl  # noqa: E402
from tests.test_hip_car_parity import make_oracle_envs, oracle_to_hip_state, push_tracks  # noqa: E402


def compare(hs, envs, stats):
    for i, e in enumerate(envs):
        touching = int(e.e["n_contact"]) > 0
        key = "touch" if touching else "free"
        for c in range(2):
            q, o = hs[i]["car"][c], e.e["car"][c]
            for f in ("cx", "cy", "a", "vx", "vy", "w"):
                got = np.concatenate([[q["hull"][f]], q["wheel"][f]]).astype(np.float64)
                want = np.concatenate([[o["hull"][f]], o["wheel"][f]]).astype(np.float64)
                err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
                stats[key]["state"] = max(stats[key]["state"], err.max())
                stats[key]["n"] += err.size
                stats[key]["exact"] += int((got == want).sum())
            for f in ("imp", "motor_imp"):
                a, b = np.asarray(q[f], np.float64), np.asarray(o[f], np.float64)
                stats[key]["imp"] = max(stats[key]["imp"], (np.abs(a - b) / np.maximum(1.0, np.abs(b))).max())
        nc = int(e.e["n_contact"])
        for k in range(nc):
            qq, oo = hs[i]["contact"][k], e.e["contact"][k]
            for f in ("nimp", "timp"):
                a, b = np.asarray(qq[f], np.float64), np.asarray(oo[f], np.float64)
                stats["touch"]["cimp"] = max(stats["touch"]["cimp"], (np.abs(a - b) / np.maximum(1.0, np.abs(b))).max())


def main():
    stats = {k: dict(state=0.0, imp=0.0, cimp=0.0, n=0, exact=0) for k in ("free", "touch")}
    n = 12
    envs = make_oracle_envs(n)
    hip = crl.HipCarVecEnv(n)
    hip.reset()
    push_tracks(hip, envs)
    rs = np.random.RandomState(4)
    for t in range(160):
        hip.set_state(oracle_to_hip_state(envs))
        acts = rs.uniform(-1, 1, (n, 2, 2)).astype(np.float32)
        if t < 60:
            acts[:, :, 1] = np.abs(acts[:, :, 1])
        hip.step_device(torch.as_tensor(acts).cuda(), render=False)
        hs = hip.get_state()
        for i, e in enumerate(envs):
            e.step(acts[i].astype(np.float64))
        compare(hs, envs, stats)
    hip.close()
    # crash scenario
    n = 8
    envs = make_oracle_envs(n, seed0=20)
    rs = np.random.RandomState(1)
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
        for k in range(30):
            e.step([[0.0, 0.0], [0.0, 0.0]])
    hip = crl.HipCarVecEnv(n)
    hip.reset()
    push_tracks(hip, envs)
    for t in range(150):
        hip.set_state(oracle_to_hip_state(envs))
        acts = np.zeros((n, 2, 2), np.float32)
        acts[:, 0, 1] = 1.0
        acts[:, 0, 0] = 0.2 * np.sin(t / 11.0)
        acts[:, 1, 1] = -0.3 if t > 90 else 0.0
        hip.step_device(torch.as_tensor(acts).cuda(), render=False)
        hs = hip.get_state()
        for i, e in enumerate(envs):
            e.step(acts[i].astype(np.float64))
        compare(hs, envs, stats)
    hip.close()
    for k, s in stats.items():
        print(k, {a: (float(b) if isinstance(b, float) else b) for a, b in s.items()}, "bit-identical fraction", s["exact"] / max(s["n"], 1))


if __name__ == "__main__":
    main()
