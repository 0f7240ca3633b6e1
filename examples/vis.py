"""The reference's vis.py (two built-in agents play N episodes on ONE cPongDouble-v0 env) on the HIP backend, call for call --
minus the window: a GPU node has no display, so the match is played without render(mode="human").

    PYTHONPATH=. python examples/vis.py --left RULE_BASED --right MEDIUM -N 3
"""
import argparse
import shutil

from competitive_rl_amd import evaluate_two_policies, get_builtin_agent_names, get_compute_action_function, make_envs

if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--left", default="RULE_BASED", type=str, help="Left agent names, must in {}.".format(get_builtin_agent_names()))
    parser.add_argument("--right", default="RULE_BASED", type=str, help="Right agent names, must in {}.".format(get_builtin_agent_names()))
    parser.add_argument("--num-episodes", "-N", default=3, type=int, help="Number of episodes to run.")
    args = parser.parse_args()

    agent_names = get_builtin_agent_names()
    print("Agent names: ", agent_names)
    print("Your chosen agents: left - {}, right - {}".format(args.left, args.right))
    assert args.left in agent_names and args.right in agent_names, agent_names

    env = make_envs("cPongDouble-v0", num_envs=1, asynchronous=False, frame_stack=None, log_dir="tmp_vis").envs[0]
    left = get_compute_action_function(args.left)
    right = get_compute_action_function(args.right)

    result = evaluate_two_policies(left, right, env=env, render=False, num_episode=args.num_episodes)
    print(result)

    env.close()
    shutil.rmtree("tmp_vis")
