"""The CPU baseline legs of bench.py on this host, side by side: the reference's SubprocVecEnv architecture (the reported value) and the same C
code as one OpenMP batch, with and without thread binding, for a few batch sizes per thread.
Usage: python tools/cpu_legs.py [fused84|raw]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[2] == "omp":  # child: one OpenMP measurement (binding is read when libgomp starts)
    import numpy as np

    from oracle import pong_oracle as po

    kind, per_thread = sys.argv[1], int(sys.argv[3])
    cores = len(os.sched_getaffinity(0))
    atlas = np.load(os.path.join(ROOT, "competitive_rl_amd", "assets", "pong_score_atlas.npz"))["atlas"]
    n = per_thread * cores
    env = po.PongOracle(n, atlas, obs_mode=po.RAW if kind == "raw" else po.GRAY, resized_dim=84, frame_stack=4 if kind == "fused84" else 1, seed=0)
    env.set_threads(cores)
    env.reset()
    acts = np.random.RandomState(0).randint(0, 3, (8, n, 2)).astype(np.int32)
    env.step(acts[0])
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < 4.0:
        env.step(acts[k % 8])
        k += 1
    dt = time.perf_counter() - t0
    print(f"  OpenMP {kind}: {per_thread} envs/thread x {cores} threads, bind={os.environ.get('OMP_PROC_BIND', '-')}: {n * k / dt:,.0f} env-steps/s ({k} steps)")
    sys.exit(0)

if __name__ == "__main__":
    kind = sys.argv[1] if len(sys.argv) > 1 else "fused84"
    from oracle import subproc_baseline as sb

    cores = len(os.sched_getaffinity(0))
    v, k, dt = sb.time_subproc("raw" if kind == "raw" else "gray_84", cores, 6.0)
    print(f"subproc architecture ({cores} one-env workers): {v:,.0f} env-steps/s")
    for bind in (None, "close"):
        for per_thread in (1, 4, 16):
            env = dict(os.environ)
            if bind:
                env.update(OMP_PROC_BIND=bind, OMP_PLACES="threads")
            subprocess.run([sys.executable, os.path.abspath(__file__), kind, "omp", str(per_thread)], env=env)
