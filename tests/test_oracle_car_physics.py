"""Known-answer checks of the oracle's Box2D 2.3 restatement (the part that cannot be pinned to
the reference because box2d-py is not installable): rigid-body integration, revolute joints
(point constraint, limits, motor torque cap), polygon mass data, sensor overlap."""
import numpy as np
import pytest

from oracle import car_oracle as co

G = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden")


def fresh_env(swap=0):
    g = np.load(G + "/car_track.npz")
    u = np.concatenate([g[f"{j}/draws"] for j in range(4)])
    e = co.CarEnv()
    assert e.reset(u, swap) > 0
    e.step(None)
    return e


def anchor_errors(e, c):
    k = co.consts()
    car = e.e["car"][c]
    h = car["hull"]
    s, cs = np.sin(np.float64(h["a"])), np.cos(np.float64(h["a"]))
    errs = []
    for w in range(4):
        an = k["anchor"][w].astype(np.float64) - k["hull_lc"].astype(np.float64)
        ax = h["cx"] + cs * an[0] - s * an[1]
        ay = h["cy"] + s * an[0] + cs * an[1]
        errs.append(np.hypot(ax - car["wheel"][w]["cx"], ay - car["wheel"][w]["cy"]))
    return np.array(errs)


def test_polygon_mass_data_matches_closed_forms():
    k = co.consts()
    # wheel box 0.56 x 1.08, density 0.1: m = rho*w*h, I = m (w^2 + h^2) / 12
    m = 0.1 * 0.56 * 1.08
    assert abs(k["wheel_mass"] - m) < 1e-7
    assert abs(k["wheel_I"] - m * (0.56 ** 2 + 1.08 ** 2) / 12) < 1e-7
    # hull: sum of 4 polygons, area by shoelace in f64
    polys = [[(-60, 130), (60, 130), (60, 110), (-60, 110)], [(-15, 120), (15, 120), (20, 20), (-20, 20)],
             [(25, 20), (50, -10), (50, -40), (20, -90), (-20, -90), (-50, -40), (-50, -10), (-25, 20)],
             [(-50, -120), (50, -120), (50, -90), (-50, -90)]]
    area = 0.0
    for p in polys:
        p = np.array(p, float) * 0.02
        area += abs(0.5 * np.sum(p[:, 0] * np.roll(p[:, 1], -1) - np.roll(p[:, 0], -1) * p[:, 1]))
    assert abs(k["hull_mass"] - area) < 1e-5


def test_car_at_rest_stays_at_rest_and_joints_hold():
    e = fresh_env()
    # Car.__init__ places the wheels at UN-rotated offsets (cd:86) while the joint anchors rotate
    # with the hull, so a freshly built car first settles onto its joints; then it must stay put
    for _ in range(30):
        e.step([[0.0, 0.0], [0.0, 0.0]])
    p0 = e.hull_position(0).copy()
    for _ in range(100):
        e.step([[0.0, 0.0], [0.0, 0.0]])
    p1 = e.hull_position(0)
    assert np.allclose(p0, p1, atol=1e-4)
    for c in range(2):
        assert anchor_errors(e, c).max() < 1e-4
        assert abs(float(e.e["car"][c]["hull"]["w"])) < 1e-5


def test_full_gas_accelerates_along_heading_without_tearing_the_joints():
    e = fresh_env()
    a0 = float(e.e["car"][0]["hull"]["a"])
    heading = np.array([-np.sin(a0), np.cos(a0)])
    p0 = e.hull_position(0)[:2].copy()
    speeds = []
    for t in range(120):
        e.step([[0.0, 1.0], [0.0, 0.0]])
        h = e.e["car"][0]["hull"]
        speeds.append(float(np.hypot(h["vx"], h["vy"])))
        assert anchor_errors(e, 0).max() < 5e-3  # b2_linearSlop = 0.005
    d = e.hull_position(0)[:2] - p0
    assert speeds[-1] > 60 and np.all(np.diff(speeds[5:]) > -1e-3)  # monotone acceleration
    assert np.dot(d, heading) / np.linalg.norm(d) > 0.999          # straight along the heading
    # momentum budget: total tyre force is capped by 4 wheels * friction limit (400 on road)
    mass = co.consts()["hull_mass"] + 4 * co.consts()["wheel_mass"]
    assert max(np.diff(speeds)) * 50 * mass <= 4 * 400.0 * 1.001


def test_steering_joint_respects_limits_and_motor_speed_cap():
    e = fresh_env()
    angles = []
    for t in range(60):
        e.step([[1.0, 0.0], [-1.0, 0.0]])  # steer(-a0): car 0 targets -1 rad (beyond the -0.4 limit)
        c0, c1 = e.e["car"][0], e.e["car"][1]
        angles.append((float(c0["wheel"][0]["a"] - c0["hull"]["a"]), float(c1["wheel"][0]["a"] - c1["hull"]["a"])))
    angles = np.array(angles)
    # joint.motorSpeed = sign * min(50*|d|, 3.0) rad/s -> at most 0.06 rad per step
    assert np.abs(np.diff(angles[:, 0])).max() <= 0.06 + 1e-4
    assert abs(angles[-1, 0] + 0.4) < 0.04 and abs(angles[-1, 1] - 0.4) < 0.04  # parked at the +-0.4 limits (+- angular slop)
    assert angles[:, 0].min() > -0.4 - 0.036 and angles[:, 1].max() < 0.4 + 0.036
    # rear wheels have no steering target other than 0
    c0 = e.e["car"][0]
    assert abs(float(c0["wheel"][2]["a"] - c0["hull"]["a"])) < 1e-3
    assert set(np.unique(e.e["car"][0]["limit_state"][:2]).tolist()) <= {0, 1}


def test_sensor_overlap_drives_tile_rewards():
    e = fresh_env()
    n = int(e.e["trk"]["n"])
    tot = np.zeros(2)
    for t in range(200):
        r, d = e.step([[0.0, 0.6], [0.0, 0.0]])
        tot += r
    v = int(e.e["tile_visited_count"][0])
    assert v >= 8
    # -0.1 per step plus 1000/n per visited tile, the last step's visits are paid one step late
    assert abs(tot[0] - (-0.1 * 200 + 1000.0 / n * v)) <= 1000.0 / n * 3 + 1e-6
    assert abs(tot[1] - (-0.1 * 200 + 1000.0 / n * int(e.e["tile_visited_count"][1]))) <= 1000.0 / n * 3 + 1e-6
    # wheels of the parked car rest on the road
    assert all(e.wheel_on_road(1, w) for w in range(4))


def test_out_of_playfield_and_step_limit_end_the_episode():
    e = fresh_env()
    e.e["car"][0]["hull"]["cx"] = 400.0  # |x| > PLAYFIELD = 333.33
    _, d = e.step([[0.0, 0.0], [0.0, 0.0]])
    assert d[0] == 1 and d[1] == 0
    e2 = fresh_env()
    e2.e["step_count"] = 1001
    _, d = e2.step([[0.0, 0.0], [0.0, 0.0]])
    assert d.tolist() == [1, 1]


def test_observation_raster_structure():
    e = fresh_env()
    for _ in range(40):
        e.step([[0.0, 0.7], [0.0, 0.7]])
    for viewer in range(2):
        img = e.render(viewer)
        assert img.shape == (96, 96)
        assert set(np.unique(img).tolist()) <= {0, 29, 44, 60, 76, 101, 103, 107, 149, 161, 176, 255}
        # own hull (gray 60 = trunc(0.299*204)) sits around the camera anchor: 16 units ahead of
        # the car is the image centre, so the car is ~28 px below it
        ys, xs = np.nonzero(img == 60)
        assert len(ys) > 10 and 66 < ys.mean() < 86 and 40 < xs.mean() < 56
        # indicator strip: rows 86.. are black except the bars
        assert (img[86:, :10] == 0).all() and (img[86:] == 0).mean() > 0.8
        # road under the car, grass somewhere
        assert np.isin(img[:86], [101, 103, 107]).mean() > 0.1 and np.isin(img[:86], [161, 176]).mean() > 0.1
    # the other car shows up in blue-gray 29 in at least one of the two views
    assert (e.render(0)[:86] == 29).any() or (e.render(1)[:86] == 29).any()


def _park_car1_ahead(e, dist):
    c0, c1 = e.e["car"][0], e.e["car"][1]
    a = float(c0["hull"]["a"])
    hd = np.array([-np.sin(a), np.cos(a)])
    off = np.array([c0["hull"]["cx"] - c1["hull"]["cx"], c0["hull"]["cy"] - c1["hull"]["cy"]]) + dist * hd
    c1["hull"]["cx"] += off[0]
    c1["hull"]["cy"] += off[1]
    for w in range(4):
        c1["wheel"][w]["cx"] += off[0]
        c1["wheel"][w]["cy"] += off[1]
    return hd


@pytest.mark.parametrize("contacts", [1, 0])
def test_rear_end_collision_pushes_instead_of_passing_through(contacts):
    e = fresh_env()
    e.e["contacts_enabled"] = contacts
    hd = _park_car1_ahead(e, 8.0)
    gaps, ncs = [], []
    for t in range(140):
        e.step([[0.0, 1.0], [0.0, 0.0]])
        c0, c1 = e.e["car"][0], e.e["car"][1]
        d = np.array([c1["hull"]["cx"] - c0["hull"]["cx"], c1["hull"]["cy"] - c0["hull"]["cy"]])
        gaps.append(float(np.dot(d, hd)))
        ncs.append(int(e.e["n_contact"]))
    if contacts:
        # hull is 5.0 long (front +2.6, rear -2.4): bumper to bumper, centres stay >= 5.0 - slop apart
        # the impact step can overlap by up to v*h (no TOI for non-bullet bodies, as in Box2D);
        # the position pass then restores the bumper-to-bumper distance
        k0 = int(np.argmax(np.array(ncs) > 0))
        assert max(ncs) >= 1 and min(gaps) > 5.0 - 0.25 and min(gaps[k0 + 10:]) > 5.0 - 0.03
        v0 = np.hypot(e.e["car"][0]["hull"]["vx"], e.e["car"][0]["hull"]["vy"])
        v1 = np.hypot(e.e["car"][1]["hull"]["vx"], e.e["car"][1]["hull"]["vy"])
        assert v1 > 10 and abs(v0 - v1) < 0.5  # the parked car is being pushed along
        k = int(np.argmax(np.array(ncs) > 0))
        assert all(n > 0 for n in ncs[k + 5:])  # resting contact persists, warm-started
        assert float(e.e["contact"][0]["nimp"].sum()) > 1.0
    else:
        assert min(gaps) < 3.0 and max(ncs) == 0  # without the contact solver the cars overlap


def test_island_sleep_zeroes_sub_tolerance_velocities_after_half_a_second():
    """b2Island::Solve's sleep rule (Box2D 2.3): bodies below the linear (0.01) / angular (2 deg/s)
    tolerances accumulate m_sleepTime; at 0.5 s (25 steps of 1/50 s, float32 accumulation) the
    island sleeps: velocities are zeroed and the timers restart.  A fast body keeps them at 0."""
    e = fresh_env()
    for _ in range(40):
        e.step([[0.0, 0.0], [0.0, 0.0]])  # settle onto the joints
    car = e.e["car"][0]
    # creep below tolerance: the timer counts up in steps of h and the velocity survives
    for b in ("hull",):
        car[b]["vx"], car[b]["vy"], car[b]["w"] = 1e-4, 0.0, 0.0
    car["wheel"]["vx"], car["wheel"]["vy"], car["wheel"]["w"] = 1e-4, 0.0, 0.0
    car["sleep_time"][:] = 0.0
    h = np.float32(1.0 / 50.0)
    acc, slept_at = np.float32(0.0), None
    for k in range(40):
        e.step([[0.0, 0.0], [0.0, 0.0]])
        acc = np.float32(acc + h)
        st = e.e["car"][0]["sleep_time"].copy()
        if slept_at is None and np.all(st == 0.0) and k > 0:
            slept_at = k
            break
        assert np.all(st == acc), (k, st, acc)
    assert slept_at is not None and 24 <= slept_at <= 26, slept_at
    c = e.e["car"][0]
    assert c["hull"]["vx"] == 0.0 and c["hull"]["vy"] == 0.0 and c["hull"]["w"] == 0.0
    assert np.all(c["wheel"]["vx"] == 0.0) and np.all(c["wheel"]["w"] == 0.0)
    # a moving car never accumulates sleep time
    for _ in range(30):
        e.step([[0.0, 1.0], [0.0, 0.0]])
    assert np.all(e.e["car"][0]["sleep_time"] == 0.0)
    assert e.e["car"][0]["hull"]["vx"] ** 2 + e.e["car"][0]["hull"]["vy"] ** 2 > 1.0


# ---------------------------------------------------------------------------------------------------------------------------
# properties that need no Box2D binary, and a count of how often the published forms of b2CollidePolygons'
# edge search disagree (DESIGN.md section 9 tabulates which Box2D revision each restated function follows)

def _momenta(e):
    """total linear momentum and angular momentum about the origin of the ten bodies, in float64"""
    k = co.consts()
    P, Lz = np.zeros(2), 0.0
    for c in range(2):
        car = e.e["car"][c]
        bodies = [(car["hull"], float(k["hull_mass"]), float(k["hull_I"]))] + [(car["wheel"][w], float(k["wheel_mass"]), float(k["wheel_I"])) for w in range(4)]
        for b, m, inertia in bodies:
            v = np.array([b["vx"], b["vy"]], np.float64)
            r = np.array([b["cx"], b["cy"]], np.float64)
            P += m * v
            Lz += inertia * float(b["w"]) + m * (r[0] * v[1] - r[1] * v[0])
    return P, Lz


def test_momentum_is_conserved_through_a_collision_without_tyre_forces():
    """Two cars, no tyre forces (world.Step alone: car_oracle_world_step), one driven into the other at an angle: every impulse the
    solver applies -- joint point / limit / motor constraints, contact normal and friction, the 2-point block solver -- acts
    equal and opposite on two bodies at one point, so the total linear momentum and the angular momentum about the origin of
    the ten bodies must survive the crash up to float32 rounding (the position pass moves bodies by at most a few slops per
    step without touching velocities: the bound on the angular part allows for that)."""
    e = fresh_env()
    e.e["contacts_enabled"] = 1
    hd = _park_car1_ahead(e, 7.0)
    for _ in range(20):
        e.step([[0.0, 0.0], [0.0, 0.0]])  # settle
    lat = np.array([hd[1], -hd[0]])
    for c, v in ((0, 9.0 * hd + 1.5 * lat), (1, -2.0 * hd)):  # car 0 into car 1, slightly across; car 1 rolls towards it
        car = e.e["car"][c]
        car["hull"]["vx"], car["hull"]["vy"] = v
        car["wheel"]["vx"], car["wheel"]["vy"] = v[0], v[1]
        car["hull"]["w"] = 0.3 if c == 0 else 0.0
        car["motor_speed"][:] = 0.0
    for c in range(2):
        for b in [e.e["car"][c]["hull"]] + [e.e["car"][c]["wheel"][w] for w in range(4)]:
            b["fx"] = b["fy"] = 0.0
    P0, L0 = _momenta(e)
    touched, worstP, worstL = 0, 0.0, 0.0
    scaleP, scaleL = np.abs(P0).max(), max(abs(L0), 1.0)
    for t in range(60):
        e.world_step()
        touched += int(e.e["n_contact"]) > 0
        P, L = _momenta(e)
        worstP, worstL = max(worstP, np.abs(P - P0).max() / scaleP), max(worstL, abs(L - L0) / scaleL)
    assert touched >= 5, touched
    assert worstP < 2e-5, worstP
    assert worstL < 2e-3, worstL
    # and the crash did exchange momentum: car 1 now moves along +hd
    assert float(np.dot([e.e["car"][1]["hull"]["vx"], e.e["car"][1]["hull"]["vy"]], hd)) > 1.0


def test_joint_anchors_are_within_linear_slop_after_every_converged_position_solve():
    """b2RevoluteJoint::SolvePositionConstraints returns `positionError <= b2_linearSlop` measured BEFORE its last correction;
    the island stops iterating when every joint says so.  While driving (no contacts) every step converges, so after the
    step each wheel's anchor is within linearSlop of the hull's anchor -- checked in float64 from the stored poses."""
    e = fresh_env()
    rs = np.random.RandomState(3)
    worst = 0.0
    for t in range(200):
        a = rs.uniform(-1, 1, (2, 2))
        a[:, 1] = abs(a[:, 1]) if t < 120 else -abs(a[:, 1])
        e.step(a)
        for c in range(2):
            worst = max(worst, float(anchor_errors(e, c).max()))
    assert worst <= 0.005 * (1 + 1e-3), worst


def test_warm_started_steady_contact_reproduces_its_impulses():
    """A pushing contact that persists: the next step's Collide finds the same manifolds with the same contact ids and hands
    the solver EXACTLY the accumulated normal / tangent impulses the previous solve ended with (b2Contact::Update copies them by
    id; b2ContactSolver::WarmStart applies them), and with that start the solve lands close to where it started: the
    impulses change by a few per cent per step instead of being rebuilt from zero."""
    e = fresh_env()
    e.e["contacts_enabled"] = 1
    _park_car1_ahead(e, 8.0)
    L = co.lib()
    carried = steady = 0
    for t in range(170):
        before = {int(e.e["contact"][k]["pair"]): e.e["contact"][k].copy() for k in range(int(e.e["n_contact"]))}
        probe = e.buf.copy()
        L.car_oracle_collide_batch(co._p(probe), 1)  # what this step's Collide will hand to the solver
        for k in range(int(probe[0]["n_contact"])):
            c = probe[0]["contact"][k]
            old = before.get(int(c["pair"]))
            if old is None:
                continue
            for i_ in range(int(c["count"])):
                match = [j_ for j_ in range(int(old["count"])) if int(old["id"][j_]) == int(c["id"][i_])]
                if match:
                    assert c["nimp"][i_] == old["nimp"][match[0]] and c["timp"][i_] == old["timp"][match[0]], (t, k, i_)
                    carried += 1
                else:
                    assert c["nimp"][i_] == 0.0 and c["timp"][i_] == 0.0, (t, k, i_)
        e.step([[0.0, 1.0], [0.0, 0.0]])
        if t >= 130 and before:
            after = {int(e.e["contact"][k]["pair"]): e.e["contact"][k] for k in range(int(e.e["n_contact"]))}
            assert set(after) == set(before), (t, sorted(before), sorted(after))
            tot_b = sum(float(c["nimp"][:int(c["count"])].sum()) for c in before.values())
            tot_a = sum(float(c["nimp"][:int(c["count"])].sum()) for c in after.values())
            assert tot_b > 1.0 and abs(tot_a - tot_b) / tot_b < 0.3, (t, tot_b, tot_a)
            steady += 1
    assert carried > 100 and steady >= 35, (carried, steady)


def test_count_where_the_published_forms_of_the_polygon_edge_search_disagree():
    """Box2D changed b2FindMaxSeparation between 2.3.0 (hill climb from the edge facing the other centroid; flip rule
    0.98 / 0.001) and 2.3.1 (exhaustive search in polygon 2's frame; flip rule + 0.1 linearSlop); the oracle and the HIP kernels
    search exhaustively with world-frame dot products and the 2.3.0 flip rule.  box2d-py ~=2.3.5 cannot
    be installed here to say which one the reference runs, so this test MEASURES the disagreement instead: a contact-rich soak
    is stepped with the default build, and at every step the three forms compute the manifolds from the same poses."""
    from tests.car_scenarios import make_oracle_envs

    n, steps, base = 192, 120, 16
    seeds = make_oracle_envs(base, seed0=40)
    B = co.CarBatch(n)
    rs = np.random.RandomState(8)
    for i in range(n):
        B.E[i] = seeds[i % base].e
    # car 1 next to car 0 at a random relative pose (some overlapping, some a little apart)
    c0, c1 = B.E["car"][:, 0], B.E["car"][:, 1]
    ang = c0["hull"]["a"].astype(np.float64) + rs.uniform(0, 2 * np.pi, n)
    dist = rs.uniform(1.8, 5.5, n)
    dx, dy = (np.cos(ang) * dist).astype(np.float32), (np.sin(ang) * dist).astype(np.float32)
    turn = rs.uniform(-np.pi, np.pi, n).astype(np.float32)
    for i in range(n):  # rotate car 1 about its hull centre by `turn`, then move it
        cs, sn = np.cos(turn[i]), np.sin(turn[i])
        hx, hy = float(c1["hull"]["cx"][i]), float(c1["hull"]["cy"][i])
        for w in range(4):
            rx, ry = float(c1["wheel"]["cx"][i, w]) - hx, float(c1["wheel"]["cy"][i, w]) - hy
            c1["wheel"]["cx"][i, w], c1["wheel"]["cy"][i, w] = hx + cs * rx - sn * ry, hy + sn * rx + cs * ry
            c1["wheel"]["a"][i, w] += turn[i]
        c1["hull"]["a"][i] += turn[i]
        ox, oy = float(c0["hull"]["cx"][i]) + dx[i] - hx, float(c0["hull"]["cy"][i]) + dy[i] - hy
        c1["hull"]["cx"][i] += ox
        c1["hull"]["cy"][i] += oy
        c1["wheel"]["cx"][i] += ox
        c1["wheel"]["cy"][i] += oy
    libs = {0: co.lib(), 1: co.collide_variant(1), 2: co.collide_variant(2)}
    tot = {k: dict(manifolds=0, missing=0, extra=0, flip=0, face=0, ids=0, points=0) for k in (1, 2)}
    ref_manifolds = 0
    for t in range(steps):
        found = {}
        for k, L in libs.items():
            Ev = B.E.copy()
            L.car_oracle_collide_batch(co._p(Ev), n)
            found[k] = Ev
        R = found[0]
        for i in range(n):
            ref = {int(R["contact"][i, j]["pair"]): R["contact"][i, j] for j in range(int(R["n_contact"][i]))}
            ref_manifolds += len(ref)
            for k in (1, 2):
                alt = {int(found[k]["contact"][i, j]["pair"]): found[k]["contact"][i, j] for j in range(int(found[k]["n_contact"][i]))}
                d = tot[k]
                d["manifolds"] += len(alt)
                d["missing"] += len(set(ref) - set(alt))
                d["extra"] += len(set(alt) - set(ref))
                for pair in set(ref) & set(alt):
                    a, b = ref[pair], alt[pair]
                    if int(a["type"]) != int(b["type"]):
                        d["flip"] += 1
                    elif tuple(a["ln"]) != tuple(b["ln"]):
                        d["face"] += 1
                    elif int(a["count"]) != int(b["count"]):
                        d["points"] += 1
                    elif not np.array_equal(a["id"][:int(a["count"])], b["id"][:int(b["count"])]):
                        d["ids"] += 1
        acts = rs.uniform(-1, 1, (n, 2, 2))
        acts[:, 1] = -acts[:, 0] if t % 20 < 10 else acts[:, 1]
        B.step(acts)
    print("manifolds (default form):", ref_manifolds, "| Box2D 2.3.1 form:", tot[1], "| Box2D 2.3.0 form:", tot[2])
    assert ref_manifolds > 3000
    for k in (1, 2):
        d = tot[k]
        differing = d["missing"] + d["extra"] + d["flip"] + d["face"] + d["ids"] + d["points"]
        # recorded in DESIGN.md section 9; the bound only keeps the statement honest if the scenario or the code changes
        assert differing <= 0.05 * ref_manifolds, (k, d, ref_manifolds)
