"""The builds of ONE oracle source that differ in the island solver's arithmetic (oracle/Makefile), CPU only:

  liboracle.so       Box2D's own roundings, shared sin / cos  == the HIP kernels' default path (tolerance 0, -m gpu tests)
  liboracle_norb.so  the same without the wheel joints' exactly-zero rB terms -- what the HIP kernels evaluate
  liboracle_fma.so   every a*b+c of the iterations in one fused multiply-add == CRL_FLAG_CAR_FMA contexts (tolerance 0, -m gpu tests)
  liboracle_libm.so  the host libm, like the reference

Box2D itself is not installable here, so the solver restatement is pinned to nothing but itself and the known-answer tests:
the digests below freeze the default build's (and the fma build's) trajectories, so a refactoring slip made identically in
the oracle and the HIP code cannot pass unnoticed (tools/oracle_regress.py prints them)."""
import hashlib

import numpy as np

from oracle import car_oracle as co
from tests.car_scenarios import crash_actions, make_oracle_envs, park_for_crash

# sha256 over car states, manifolds and rewards of tools/oracle_regress.py's two soaks, recorded from the round-4 oracle (before the
# MAD / NMAD sites went in) for the default and libm builds, and from the first fma build
DIGEST = {
    False: "a89d20498c82ba82fdd4f4caed17104e107d0f2d4f998e07ca15b21a87d11bd1",
    "norb": "a89d20498c82ba82fdd4f4caed17104e107d0f2d4f998e07ca15b21a87d11bd1",
    "fma": "0ec3ce5be4f30b3b0bd6903848014b44c0860150e189c7b008ab749ce098d12c",
}


def _soak_digest(variant):
    h = hashlib.sha256()

    def eat(envs):
        for e in envs:
            h.update(e.e["car"].tobytes())
            h.update(e.e["contact"].tobytes())
            h.update(np.int32(e.e["n_contact"]).tobytes())
            h.update(e.e["reward"].tobytes())

    n = 12
    envs = make_oracle_envs(n, libm=variant)
    rs = np.random.RandomState(4)
    for t in range(200):
        acts = rs.uniform(-1, 1, (n, 2, 2)).astype(np.float32)
        if t < 60:
            acts[:, :, 1] = np.abs(acts[:, :, 1])
        for e, a in zip(envs, acts):
            e.step(a.astype(np.float64))
        eat(envs)
    n = 16
    envs = make_oracle_envs(n, seed0=20, libm=variant)
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
    return h.hexdigest(), touched


def test_default_build_trajectories_are_frozen():
    d, touched = _soak_digest(False)
    assert touched > 2000
    assert d == DIGEST[False]


def test_dropping_the_zero_rb_terms_changes_no_bit():
    """A wheel's joint anchor is its centre (rB = 0): wB x rB and rB x P add +-0.  The HIP kernels do not evaluate those terms; the
    oracle's `norb` build (the default build without them) walks the same 5 800 env-steps, 2 598 of them touching, to the same bytes."""
    d, _ = _soak_digest("norb")
    assert d == DIGEST[False]


def test_fma_build_trajectories_are_frozen():
    d, touched = _soak_digest("fma")
    assert touched > 2000
    assert d == DIGEST["fma"] and d != DIGEST[False]
