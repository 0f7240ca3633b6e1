"""Longer teacher-forced CarRacing parity run (one-off, GPU box): every step starts from the oracle's state,
both sides step once, bodies / joints / sleep timers / tile bookkeeping are compared.
PYTHONPATH=. python tools/car_teacher_soak.py [envs] [steps]"""
import sys

import numpy as np
import torch

sys.path.insert(0, "tests")
import competitive_rl_amd as crl
from test_hip_car_parity import make_oracle_envs, oracle_to_hip_state, push_tracks

n, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16, int(sys.argv[2]) if len(sys.argv) > 2 else 600
envs = make_oracle_envs(n, seed0=5)
hip = crl.HipCarVecEnv(n)
hip.reset()
push_tracks(hip, envs)
rs = np.random.RandomState(12)
worst, worst_touch, touching = 0.0, 0.0, 0
alive = [True] * n  # an env that finished is auto-reset on the HIP side only: it leaves the comparison
for t in range(steps):
    hip.set_state(oracle_to_hip_state(envs))
    acts = rs.uniform(-1, 1, (n, 2, 2)).astype(np.float32)
    if (t // 50) % 2 == 0:
        acts[:, :, 1] = np.abs(acts[:, :, 1])
    if t % 97 < 30:
        acts[:, 1] = acts[:, 0]  # both cars do the same: they stay close and bump into each other
    hip.step_device(torch.as_tensor(acts).cuda(), render=False)
    hs = hip.get_state()
    for i, e in enumerate(envs):
        _, dn = e.step(acts[i].astype(np.float64))
        if not alive[i]:
            continue
        if any(dn) or t >= 998:
            alive[i] = False
            continue
        touch = int(e.e["n_contact"]) > 0
        touching += touch
        for c in range(2):
            q, o = hs[i]["car"][c], e.e["car"][c]
            for f in ("cx", "cy", "a", "vx", "vy", "w"):
                for got, want in ((q["hull"][f], o["hull"][f]), *zip(q["wheel"][f], o["wheel"][f])):
                    err = abs(float(got) - float(want)) / max(1.0, abs(float(want)))
                    if touch:
                        worst_touch = max(worst_touch, err)
                    else:
                        worst = max(worst, err)
            assert np.array_equal(q["sleep_time"], o["sleep_time"]), (t, i, c)
            assert np.array_equal(q["limit_state"], o["limit_state"]), (t, i, c)
            if int(q["tile_visited_count"]) != int(e.e["tile_visited_count"][c]):
                hv, ov = np.asarray(q["visited"]), np.asarray(e.e["visited"][c])
                diff = [(w * 32 + b) for w in range(len(hv)) for b in range(32) if ((int(hv[w]) ^ int(ov[w])) >> b) & 1]
                print("t", t, "env", i, "car", c, "visited diff tiles", diff, "hip count", int(q["tile_visited_count"]), "oracle", int(e.e["tile_visited_count"][c]))
                print("  hip wheel_tiles", [[k for k in range(512) if (int(q["wheel_tiles"][w][k >> 5]) >> (k & 31)) & 1] for w in range(4)])
                print("  ora wheel_tiles", [[k for k in range(512) if (int(e.e["wheel_tiles"][c][w][k >> 5]) >> (k & 31)) & 1] for w in range(4)])
                print("  ntiles", int(e.e["trk"]["n"]), "last_block hip/oracle", int(q["last_block"]), int(e.e["last_block"][c]))
            assert int(q["tile_visited_count"]) == int(e.e["tile_visited_count"][c]), (t, i, c)
            assert int(q["done"]) == int(e.e["done"][c]), (t, i, c)
        assert int(hs[i]["n_contact"]) == int(e.e["n_contact"]), (t, i)
    if not (worst < 1e-5 and worst_touch < 5e-3):
        for i, e in enumerate(envs):
            nc = int(e.e["n_contact"])
            if nc == 0:
                continue
            for c in range(2):
                q, o = hs[i]["car"][c], e.e["car"][c]
                d = max(abs(float(q["hull"][f]) - float(o["hull"][f])) for f in ("cx", "cy", "a", "vx", "vy", "w"))
                dw = max(float(np.abs(q["wheel"][f] - o["wheel"][f]).max()) for f in ("cx", "cy", "a", "vx", "vy", "w"))
                print("env", i, "car", c, "nc", nc, "hull err", d, "wheel err", dw)
            for k in range(nc):
                ho, oo = hs[i]["contact"][k], e.e["contact"][k]
                print("  contact", k, "oracle pair/count/type", int(oo["pair"]), int(oo["count"]), int(oo["type"]), "nimp", oo["nimp"], "timp", oo["timp"], "id", oo["id"])
                print("           hip    pair/count/type", int(ho["pair"]), int(ho["count"]), int(ho["type"]), "nimp", ho["nimp"], "timp", ho["timp"], "id", ho["id"])
    # while touching, a wheel in contact turns 1e-5 of relative impulse difference into ~1e-3 of its angular
    # velocity (inverse inertia 134): the bar there is on that scale
    assert worst < 1e-5 and worst_touch < 5e-3, (t, worst, worst_touch)
print("steps", steps, "envs", n, "still compared at the end", sum(alive), "touching env-steps", touching, "worst relative state error: free", worst, "touching", worst_touch)
hip.close()
