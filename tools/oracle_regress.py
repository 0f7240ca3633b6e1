#!/usr/bin/env python3
"""Digest of the oracle's CarRacing state over a free-driving and a crash soak (CPU only).

Used when the oracle's solver source is restructured (round 5: the MAD / NMAD sites of the -DCRL_FMA build): the digest of
the DEFAULT build must not change -- the Box2D part of the oracle is pinned to nothing but itself and the known-answer
tests, so a slip made identically in oracle and HIP would pass every HIP-vs-oracle test.

    python tools/oracle_regress.py [variant]      variant: "" (liboracle.so), "libm", "fma"
"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import car_oracle as co  # noqa: E402
from tests.car_scenarios import crash_actions, make_oracle_envs, park_for_crash  # noqa: E402


def main():
    variant = sys.argv[1] if len(sys.argv) > 1 else ""
    libm = {"": False, "libm": True}.get(variant, variant)
    h = hashlib.sha256()

    def eat(envs):
        for e in envs:
            h.update(e.e["car"].tobytes())
            h.update(e.e["contact"].tobytes())
            h.update(np.int32(e.e["n_contact"]).tobytes())
            h.update(e.e["reward"].tobytes())

    n = 12
    envs = make_oracle_envs(n, libm=libm)
    rs = np.random.RandomState(4)
    for t in range(200):
        acts = rs.uniform(-1, 1, (n, 2, 2)).astype(np.float32)
        if t < 60:
            acts[:, :, 1] = np.abs(acts[:, :, 1])
        for e, a in zip(envs, acts):
            e.step(a.astype(np.float64))
        eat(envs)
    n = 16
    envs = make_oracle_envs(n, seed0=20, libm=libm)
    park_for_crash(envs)
    touched = 0
    for t in range(260):
        acts = crash_actions(n, t)
        if t > 150:
            acts[:, 1, 0] = 0.8 * np.sin(t / 5.0 + np.arange(n))
            acts[:, 0, 0] = -0.9 * np.cos(t / 7.0 + np.arange(n))
        for e, a in zip(envs, acts):
            e.step(a.astype(np.float64))
            touched += int(e.e["n_contact"]) > 0
        eat(envs)
    print("variant=%r touched env-steps=%d digest=%s" % (variant, touched, h.hexdigest()))


if __name__ == "__main__":
    main()
