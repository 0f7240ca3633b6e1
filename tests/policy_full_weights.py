"""Seeded weights for the full-size ActorCritic (reference utils/network.py:14-70).  The reference tree ships no checkpoint of that
network (STRONG / ALPHA_PONG), so the golden vectors of tests/golden/policy_full.npz pin its FORWARD PASS on these weights: the
generator loads them into the reference's own torch module, the tests regenerate the same arrays from the seed."""
import numpy as np

SHAPES = {"conv1_w": (16, 4, 4, 4), "conv1_b": (16,), "conv2_w": (32, 16, 4, 4), "conv2_b": (32,), "conv3_w": (256, 32, 11, 11),
          "conv3_b": (256,), "actor_w": (3, 256), "actor_b": (3,), "critic_w": (1, 256), "critic_b": (1,)}


def make_weights(seed=2024):
    rs = np.random.RandomState(seed)
    w = {}
    for k, shp in SHAPES.items():
        if k.endswith("_w"):
            fan_in = int(np.prod(shp[1:]))
            w[k] = (rs.standard_normal(shp) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        else:
            w[k] = (rs.standard_normal(shp) * 0.1).astype(np.float32)
    return w


def make_stacks(seed=7, count=12):
    """u8 stacks [count, 4, 42, 42]: dense noise and sparse Pong-like planes"""
    rs = np.random.RandomState(seed)
    x = rs.randint(0, 256, (count, 4, 42, 42)).astype(np.uint8)
    for i in range(count // 3):
        x[i] = (rs.random_sample((4, 42, 42)) > 0.9).astype(np.uint8) * 255
    return x
