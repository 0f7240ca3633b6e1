"""CPU experiment (round 5): how many position iterations does a car ON ITS OWN need?  (`make -C oracle cyc` build; the oracle counts them.)
99.7 % converge in one, 99.9 % within three, 0.09 % run all 60 -- so one per-car-solve wavefront (64 cars) in twenty runs 60 iterations for one lane.

    python tools/lone_pos_hist.py
"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import car_oracle as co, pong_oracle as po
import subprocess
subprocess.check_call(["make", "-C", po.HERE, "cyc"], stdout=subprocess.DEVNULL)
L = co._configure(C.CDLL(os.path.join(po.HERE, "liboracle_cyc.so")))
L.car_oracle_lone_pos_hist.argtypes = [C.c_void_p]
n, steps = 256, 600
B = co.CarBatch(n); B.L = L
rs = np.random.RandomState(5)
def reset(i):
    v = B.view(i); v.L = L
    while True:
        u = rs.random_sample(24 * 8)
        if v.reset(u, int(rs.randint(0, 2))) > 0: break
    B.E[i]["contacts_enabled"] = 1
    v.step(None)
for i in range(n): reset(i)
acts = rs.uniform(-1, 1, (16, n, 2, 2))
for t in range(steps):
    _, d = B.step(acts[t % 16])
    for i in np.nonzero(d.any(1))[0]: reset(int(i))
h = np.zeros(64, np.int64); L.car_oracle_lone_pos_hist(h.ctypes.data_as(C.c_void_p))
tot = h[1:61].sum()
print("lone islands", tot, "unsolved", h[0], "with a limit active", h[61])
print("hist 1..10", h[1:11].tolist()); print("11..59", h[11:60].sum(), "60", h[60])
cum = np.cumsum(h[1:61]) / tot
for k in (1,2,3,4,5,8,10,20,40,59): print(k, f"{cum[k-1]:.5f}")
