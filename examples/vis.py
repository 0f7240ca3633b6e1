"""Two built-in Pong agents play a few matches on ONE cPongDouble-v0 env of the HIP backend and the win / draw / loss books are printed --
the job of the reference's top-level vis.py (same command line: --left, --right, -N), without its window: a GPU node has no display.

    PYTHONPATH=. python examples/vis.py --left RULE_BASED --right MEDIUM -N 3
"""
import argparse
import tempfile

import competitive_rl_amd as crl


def play(left_name, right_name, episodes):
    names = crl.get_builtin_agent_names()
    for side, name in (("left", left_name), ("right", right_name)):
        if name not in names:
            raise SystemExit(f"--{side} {name}: pick one of {names}")
    with tempfile.TemporaryDirectory(prefix="vis_") as log_dir:
        batch_of_one = crl.make_envs("cPongDouble-v0", num_envs=1, asynchronous=False, frame_stack=None, log_dir=log_dir)
        single_env = batch_of_one.envs[0]          # gym's reset() / step() on the one env, as vis.py uses it
        books = crl.evaluate_two_policies(crl.get_compute_action_function(left_name), crl.get_compute_action_function(right_name),
                                          env=single_env, num_episode=episodes, render=False)
        single_env.close()
        batch_of_one.close()
    return books


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("--left", default="RULE_BASED", help="agent on the left bat")
    ap.add_argument("--right", default="RULE_BASED", help="agent on the right bat")
    ap.add_argument("--num-episodes", "-N", type=int, default=3, help="matches to play")
    a = ap.parse_args()
    print(f"agents: {crl.get_builtin_agent_names()}; left = {a.left}, right = {a.right}")
    print(play(a.left, a.right, a.num_episodes))
