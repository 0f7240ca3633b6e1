"""Golden vectors for the FULL-SIZE opponent network (reference utils/network.py:14-70, ActorCritic: conv 4->16 k4 s2, conv 16->32 k4
s2 pad 2, conv 32->256 k11, linear 256->3 / 256->1) -- the model STRONG and ALPHA_PONG use (pong/builtin_policies.py:63-80).

Run in the build container only (needs /root/reference and torch):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_policy_full_golden.py

The reference tree ships no checkpoint of this network, so the vectors pin its forward pass on seeded weights
(tests/policy_full_weights.py): the weights are loaded into the reference's own torch module through ``load_state_dict`` and the
module is called on uint8 stacks exactly as Policy.compute_action does (policy_serving.py:49-58: ``self.model(obs)``; the
division by 255 is inside forward).  Recorded: ``logits`` f32 [B, 3], ``values`` f32 [B], the seeds."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _ref_standins as S  # noqa: E402
from tests.policy_full_weights import make_stacks, make_weights  # noqa: E402


def main():
    S.install()
    net = S.load_ref("competitive_rl.utils.network", "utils/network.py")
    torch.set_num_threads(1)
    wseed, xseed = 2024, 7
    w, x = make_weights(wseed), make_stacks(xseed)
    model = net.ActorCritic((4, 42, 42), 3)
    sd = {"conv1.weight": w["conv1_w"], "conv1.bias": w["conv1_b"], "conv2.weight": w["conv2_w"], "conv2.bias": w["conv2_b"],
          "conv3.weight": w["conv3_w"], "conv3.bias": w["conv3_b"], "actor_linear.weight": w["actor_w"], "actor_linear.bias": w["actor_b"],
          "critic_linear.weight": w["critic_w"], "critic_linear.bias": w["critic_b"]}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.eval()
    with torch.no_grad():
        logits, values = model(torch.from_numpy(x).float())
    out = os.path.join(HERE, "policy_full.npz")
    np.savez_compressed(out, weight_seed=wseed, stack_seed=xseed, logits=logits.numpy().astype(np.float32),
                        values=values.numpy().reshape(-1).astype(np.float32), feature_size=np.int64(model.feature_size((4, 42, 42))))
    print("wrote", out, "logits", logits.numpy()[:3], "feature size", model.feature_size((4, 42, 42)))


if __name__ == "__main__":
    main()
