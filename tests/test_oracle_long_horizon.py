"""Long free-running trajectories are chaotic: after a few hundred steps a last-bit difference of one sine has grown to a
different trajectory, so the builds of the oracle cannot be compared state by state over an episode.
What has to hold instead is that they are the same SIMULATOR statistically.  CPU only, whole episodes:

  liboracle_libm.so   what the reference's Box2D / CPython compute (host libm)            -- the yardstick
  liboracle.so        shared sin / cos  (== the HIP kernels' default path, tolerance 0)
  liboracle_fma.so    + the island iterations in fused multiply-adds (== CRL_FLAG_CAR_FMA contexts, tolerance 0)

512 envs x one full episode each (TimeLimit 1000 or an earlier `done`), the SAME tracks and the SAME action stream per env in
every build; per env: episode return of both cars, tiles visited, episode length, steps with car-car contact.  Compared with a
two-sample Kolmogorov-Smirnov test and a paired bootstrap of the mean difference; the numbers are printed and quoted in
DESIGN.md section 6."""
import numpy as np
import pytest

from oracle import car_oracle as co

N_ENVS = 512
MAX_STEPS = 1000


def _episodes(variant, n=N_ENVS, seed=11):
    """one episode per env; returns dict of per-env arrays"""
    rs = np.random.RandomState(seed)
    B = co.CarBatch(n, libm=variant)
    for i in range(n):
        v = B.view(i)
        for _ in range(8):
            if v.reset(rs.random_sample(24 * 8), int(rs.randint(0, 2))) > 0:
                break
        else:
            raise AssertionError("no track")
        B.E[i]["contacts_enabled"] = 1
        v.step(None)
    # the action stream: smooth steering (an AR(1) process per car), mostly on the gas, now and then the brake -- the same in every build.
    # Car 1 follows car 0's steering with a lag for the first third of the envs: the cars stay close and touch.
    steer = np.zeros((n, 2))
    ret = np.zeros((n, 2))
    length = np.zeros(n, np.int64)
    contact_steps = np.zeros(n, np.int64)
    tiles = np.zeros((n, 2), np.int64)
    alive = np.ones(n, bool)
    follow = np.arange(n) % 3 == 0
    for t in range(MAX_STEPS + 1):
        steer = 0.9 * steer + 0.35 * rs.standard_normal((n, 2))
        gas = np.where(rs.random_sample((n, 2)) < 0.08, -rs.random_sample((n, 2)), 0.2 + 0.8 * rs.random_sample((n, 2)))
        a = np.stack([np.clip(steer, -1, 1), gas], 2)
        a[follow, 1, 0] = a[follow, 0, 0]
        a[follow, 1, 1] = np.maximum(a[follow, 0, 1], 0.3) + 0.1
        r, d = B.step(np.clip(a, -1, 1))
        ret[alive] += r[alive]
        length[alive] += 1
        contact_steps[alive] += (B.E["n_contact"][alive] > 0)
        tiles[alive] = B.E["tile_visited_count"][alive]
        done = d.any(1) | (B.E["step_count"] >= MAX_STEPS)
        alive &= ~done
        if not alive.any():
            break
    assert not alive.any()
    return dict(ret0=ret[:, 0], ret1=ret[:, 1], tiles=tiles.sum(1).astype(float), length=length.astype(float), contact=contact_steps.astype(float))


@pytest.fixture(scope="module")
def episodes():
    return {k: _episodes(v) for k, v in (("libm", True), ("crl", False), ("fma", "fma"))}


def _compare(a, b, label):
    from scipy.stats import ks_2samp

    rs = np.random.RandomState(0)
    out = {}
    for k in a:
        ks = ks_2samp(a[k], b[k], method="asymp")
        d = b[k] - a[k]  # paired: same env, same track, same actions
        boot = np.array([d[rs.randint(0, len(d), len(d))].mean() for _ in range(2000)])
        lo, hi = np.percentile(boot, [0.5, 99.5])
        out[k] = dict(ks=float(ks.statistic), p=float(ks.pvalue), mean_a=float(a[k].mean()), mean_b=float(b[k].mean()), ci=(float(lo), float(hi)),
                      identical=float((d == 0).mean()), sd=float(a[k].std()))
    print(label)
    for k, v in out.items():
        print(f"   {k:8s} mean {v['mean_a']:10.3f} vs {v['mean_b']:10.3f}  (sd {v['sd']:8.3f})  KS D={v['ks']:.4f} p={v['p']:.3f}  "
              f"99 % CI of the paired mean difference [{v['ci'][0]:+.3f}, {v['ci'][1]:+.3f}]  envs with the identical value {100 * v['identical']:.1f} %")
    return out


@pytest.mark.parametrize("build", ["crl", "fma"])
def test_whole_episodes_are_statistically_the_reference_simulator(episodes, build):
    a, b = episodes["libm"], episodes[build]
    assert a["contact"].sum() > 2000 and (a["tiles"] > 20).mean() > 0.5, "the workload must drive and collide"
    out = _compare(a, b, f"{N_ENVS} whole episodes, liboracle_libm.so vs {'liboracle.so' if build == 'crl' else 'liboracle_fma.so'}:")
    for k, v in out.items():
        # the same distribution (KS cannot tell them apart at the 1 % level) and no shift of the mean beyond its sampling noise
        assert v["p"] > 0.01, (k, v)
        assert v["ci"][0] <= 0.0 <= v["ci"][1] or abs(v["mean_b"] - v["mean_a"]) <= 0.02 * max(v["sd"], 1e-9), (k, v)
