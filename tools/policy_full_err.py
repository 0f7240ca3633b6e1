import sys, numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0] if "/" in __file__ else "..")
from tests.policy_full_weights import make_weights
from tests.test_hip_policy_full import make_policy
from oracle import policy_oracle as P
n = 96
w = make_weights(11)
pol, ora = make_policy(n, w), P.PolicyOracle(w, n, full=True)
rs = np.random.RandomState(0)
err = 0
for t in range(6):
    f = rs.randint(0, 256, (n, 1, 42, 42)).astype(np.uint8)
    pol.act_device(torch.from_numpy(f).cuda(), want_logits=True)
    ora(f)
    err = max(err, float(np.abs(pol.logits().cpu().numpy() - ora.logits).max()))
print("max |logit - oracle| =", err, " logit scale", float(np.abs(ora.logits).max()))
