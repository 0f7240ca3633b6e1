#!/usr/bin/env python3
"""CPU experiment (VERDICT r04 #1b): do the velocity / position iterations of TOUCHING islands enter a bit-exact cycle of
period <= 8?  If the slow islands did, the remaining iterations could be skipped exactly.

Runs the oracle build with -DCRL_CYCLE_STATS (`make -C oracle cyc`) over the bench's kind of workload: n envs, 16 cycled
action tensors (cars circle and keep meeting), episodes restarted on done.

    python tools/cycle_stats.py [fma] [n] [steps]
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import car_oracle as co  # noqa: E402
from oracle import pong_oracle as po  # noqa: E402


def main():
    fma = len(sys.argv) > 1 and sys.argv[1] == "fma"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1200
    subprocess.check_call(["make", "-C", po.HERE, "cyc"], stdout=subprocess.DEVNULL)
    L = co._configure(C.CDLL(os.path.join(po.HERE, "liboracle_fma_cyc.so" if fma else "liboracle_cyc.so")))
    L.car_oracle_cycle_stats.argtypes = [C.c_void_p]
    B = co.CarBatch(n)
    B.L = L
    rs = np.random.RandomState(5)

    def reset(i):
        v = B.view(i)
        v.L = L
        while True:
            u = rs.random_sample(24 * 8)
            att = v.reset(u, int(rs.randint(0, 2)))
            if att > 0:
                break
        B.E[i]["contacts_enabled"] = 1
        v.step(None)

    for i in range(n):
        reset(i)
    acts = rs.uniform(-1, 1, (16, n, 2, 2))
    touching = 0
    for t in range(steps):
        _, d = B.step(acts[t % 16])
        touching += int((B.E["n_contact"] > 0).sum())
        for i in np.nonzero(d.any(1))[0]:
            reset(int(i))
    st = np.zeros(256, np.int64)
    L.car_oracle_cycle_stats(st.ctypes.data_as(C.c_void_p))
    print(f"build={'fma' if fma else 'default'} envs={n} steps={steps} touching env-steps={touching}")
    for blk, label in ((0, "all touching islands"), (100, "islands with >= 2 manifolds")):
        s = st[blk:blk + 100]
        isl = int(s[0])
        if not isl:
            continue
        vel = s[2:10]
        print(f"-- {label}: {isl}")
        print(f"   velocity iterations: cycle of period p found in {int(vel.sum())} ({100.0 * vel.sum() / isl:.1f} %), by p=1..8: {vel.tolist()},"
              f" mean iteration where first seen {s[10] / max(vel.sum(), 1):.1f}")
        hist = s[23:83]
        full = int(s[11])
        print(f"   position iterations: converged within 1-5: {int(hist[:5].sum())}, 6-20: {int(hist[5:20].sum())}, 21-59: {int(hist[20:59].sum())},"
              f" ran all 60 without converging: {full} ({100.0 * full / isl:.1f} %)")
        pc = s[13:21]
        print(f"   of those {full}: cycle of period p found in {int(pc.sum())}, by p=1..8: {pc.tolist()}, mean iteration where first seen {s[21] / max(pc.sum(), 1):.1f}")
    print("   islands by manifold count 0..8:", st[90:99].tolist())


if __name__ == "__main__":
    main()
