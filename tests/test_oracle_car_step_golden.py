"""CarRacing.step bookkeeping: the oracle against rows recorded from the reference's own
``CarRacing.step`` driven with a scripted Box2D stand-in (tests/golden/gen_car_step_golden.py): time penalty,
step-reward delta before the world step, the three done rules, done cars skipped, num_steps, action repeat."""
import numpy as np
import pytest

from tests import car_books as cb


@pytest.fixture(scope="module")
def g():
    return cb.load()


def test_fixture_covers_the_rules(g):
    sc = g["scenario"]
    assert set(sc.tolist()) == {"lap", "out", "timeout", "repeat4_timeout", "repeat2_out", "single"}
    lap = sc == "lap"
    assert (g["pre_visited_count"][lap][:, 0] == g["ntiles"][lap]).any()          # all tiles visited -> done
    assert (np.abs(g["pre_pos"][sc == "out"]) > 2000 / 6.0).any()                   # left the playfield
    assert (g["pre_step_count"][sc == "timeout"] > 1000).any()                      # step_count rule
    assert g["rew"][lap].max() > 3.0 and np.isclose(g["rew"][sc == "timeout"].min(), -0.1)
    r4 = sc == "repeat4_timeout"
    assert np.unique(np.round(g["rew"][r4][:, 0], 6)).size >= 3                      # -0.1, a partial step, 0.0


def test_oracle_step_bookkeeping_matches_reference(g):
    n = int(g["count"])
    checked_done, checked_partial = 0, 0
    for r in range(n):
        if not cb.usable(g, r):
            continue
        e = cb.env_for_row(g, r)
        players, rep = int(g["players"][r]), int(g["repeat"][r])
        rew, done = e.step_repeat(g["action"][r], rep)
        for c in range(players):
            assert rew[c] == g["rew"][r][c], (r, c, rew[c], g["rew"][r][c])      # f64, bit for bit
            assert done[c] == g["done_out"][r][c], (r, c)
            # after the call: prev_reward caught up with reward, unless the car was skipped
            assert e.e["prev_reward"][c] == g["post_prev_reward"][r][c], (r, c)
        assert int(e.e["step_count"]) == int(g["num_steps"][r]) == int(g["post_step_count"][r]), r
        checked_done += int(done[:players].any())
        checked_partial += int(rep > 1 and 0 < -rew[0] < 0.0999)
    assert checked_done > 100 and checked_partial >= 1
