"""Golden vectors for the built-in CNN opponents: the REFERENCE's own code and weights.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_policy_golden.py

Loads, by path, the reference's ``utils/network.py`` (LightActorCritic), ``utils/utils.py``
(FrameStackTensor) and ``utils/policy_serving.py`` (Policy), builds ``Policy(Box(1,42,42),
Discrete(3), N, resources/pong/checkpoint-<name>.pkl, use_light_model=True)`` exactly as
pong/builtin_policies.py:61-91 does for WEAK and MEDIUM, and drives it in closed loop as
pong/competitive_pong_env.py:36-44 does: the opponent sees ``obs[1]`` of the previous step and
plays the right bat.  The Pong frames come from this build's CPU oracle (42x42, no stack), the left
bat plays seeded random actions.

Recorded per opponent: ``frames`` u8 [T+1, N, 42, 42] (obs[1] handed to the policy at each call),
``actions`` i64 [T+1, N] (what Policy.__call__ returned), ``logits`` f32 [T+1, N, 3] and ``values``
f32 [T+1, N] (the model outputs on the policy's own 4-frame stack), plus ``noise`` / ``noise_logits``:
random u8 stacks [B, 4, 42, 42] through the bare network.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _ref_standins as S  # noqa: E402


def main():
    S.install()
    import gym

    net = S.load_ref("competitive_rl.utils.network", "utils/network.py")
    ut = S.load_ref("competitive_rl.utils.utils", "utils/utils.py")
    pkg = sys.modules["competitive_rl.utils"]
    pkg.LightActorCritic, pkg.ActorCritic, pkg.FrameStackTensor = net.LightActorCritic, net.ActorCritic, ut.FrameStackTensor
    ps = S.load_ref("competitive_rl.utils.policy_serving", "utils/policy_serving.py")

    from oracle import pong_oracle as po

    atlas = np.load(os.path.join(ROOT, "competitive_rl_amd", "assets", "pong_score_atlas.npz"))["atlas"]
    N, T = 6, 400
    torch.manual_seed(0)
    torch.set_num_threads(1)
    out = {}
    for name in ("weak", "medium"):
        pol = ps.Policy(gym.spaces.Box(0, 255, (1, 42, 42)), gym.spaces.Discrete(3), N,
                        "/root/reference/resources/pong/checkpoint-%s.pkl" % name, use_light_model=True)
        env = po.PongOracle(N, atlas, obs_mode=po.GRAY, resized_dim=42, frame_stack=1, seed=77)
        left = np.random.RandomState(3).randint(0, 3, (T, N))
        frames = np.zeros((T + 1, N, 42, 42), np.uint8)
        actions = np.zeros((T + 1, N), np.int64)
        logits = np.zeros((T + 1, N, 3), np.float32)
        values = np.zeros((T + 1, N), np.float32)
        dones = np.zeros((T, N), np.uint8)
        obs = env.reset().copy()
        for t in range(T + 1):
            frames[t] = obs[:, 1, 0]
            a = pol(obs[:, 1].copy())  # (N, 1) int64 ndarray; updates the policy's own frame stack
            actions[t] = np.asarray(a).reshape(-1)
            with torch.no_grad():
                lg, v = pol.model(pol.frame_stack.get())
            logits[t], values[t] = lg.numpy(), v.numpy().reshape(-1)
            assert np.array_equal(lg.argmax(1).numpy(), actions[t])
            if t == T:
                break
            obs, _, d = env.step(np.stack([left[t], actions[t]], 1))
            obs, dones[t] = obs.copy(), d
        env.close()
        rs = np.random.RandomState(11)
        noise = rs.randint(0, 256, (16, 4, 42, 42)).astype(np.uint8)
        noise[:4] = rs.randint(0, 2, (4, 4, 42, 42)) * 255
        with torch.no_grad():
            nl, nv = pol.model(torch.from_numpy(noise.astype(np.float32)))
        out.update({name + "_frames": frames, name + "_actions": actions, name + "_logits": logits, name + "_values": values,
                    name + "_dones": dones, name + "_left": left, name + "_noise_logits": nl.numpy(),
                    name + "_noise_values": nv.numpy().reshape(-1)})
        out["noise"] = noise
        print(name, "actions histogram", np.bincount(actions.reshape(-1), minlength=3), "episodes ended", int(dones.sum()))
    path = os.path.join(HERE, "policy_light.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
