"""CarRacing.step bookkeeping through the HIP path: every row recorded from the reference's own
``CarRacing.step`` (tests/golden/car_step_books.npz, scripted Box2D stand-in) becomes one env of a batch, teacher-forced
to the row's pre-step state; ONE ``crl_step`` must return the recorded step rewards (bit for bit, as float32), per-car
done flags and ``num_steps``."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_hip_step_bookkeeping_matches_reference():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    import competitive_rl_amd as crl
    from tests import car_books as cb
    from tests.test_hip_car_parity import oracle_to_hip_state, push_tracks

    g = cb.load()
    rows = [r for r in range(int(g["count"])) if cb.usable(g, r)]
    groups = {}
    for r in rows:
        groups.setdefault((int(g["players"][r]), int(g["repeat"][r])), []).append(r)
    assert set(groups) == {(2, 1), (2, 4), (2, 2), (1, 1)}
    checked = 0
    for (players, rep), rs in groups.items():
        n = len(rs)
        envs = [cb.env_for_row(g, r) for r in rs]
        hip = crl.HipCarVecEnv(n, players=players, action_repeat=rep, car_contacts=False)
        hip.reset()
        push_tracks(hip, envs)
        for e in envs:  # the HIP state keeps at most 6 touched tiles per wheel (a 0.6 x 1.1 wheel overlaps two or three tiles; the library
            wt = e.e["wheel_tiles"]  # COUNTS any overflow, crl_car_cap_hits, and the full-size tests assert zero); the scripted engine of
            # the fixture piled up dozens on one wheel -- that only feeds the friction limit, not the bookkeeping under test
            for c in range(2):
                for w in range(4):
                    bits = np.flatnonzero(np.unpackbits(wt[c, w].view(np.uint8), bitorder="little"))
                    for b in bits[6:]:
                        wt[c, w, b >> 5] &= ~np.uint32(1 << (b & 31))
        st = oracle_to_hip_state(envs)
        st["elapsed"] = 0
        hip.set_state(st)
        acts = np.stack([g["action"][r][:players] for r in rs]).astype(np.float32)
        _, rew, done = hip.step_device(torch.as_tensor(acts).cuda())
        dc, ns = (x.cpu().numpy() for x in hip._info_snapshot())
        rew, done = rew.cpu().numpy(), done.cpu().numpy()
        for j, r in enumerate(rs):
            want = g["rew"][r][:players].astype(np.float32)
            assert np.array_equal(rew[j], want), (players, rep, r, rew[j], want)
            assert np.array_equal(dc[j], g["done_out"][r][:players]), (players, rep, r)
            assert ns[j] == g["num_steps"][r], (players, rep, r)
            assert done[j] == int(g["done_out"][r][:players].any()), (players, rep, r)   # FlattenMultiAgentObservation: any
            checked += 1
        hip.close()
    assert checked == len(rows) > 1900
