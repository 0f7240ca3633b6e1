#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the cPongDouble hot path on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload raw|fused84|fused84_newest|car|tournament] [--gather none|scalars|obs]

A "step" is one VecEnv.step over this rank's shard of envs with synthetic (pre-generated,
device-resident) random actions, auto-reset included, no host sync inside the timed loop.
Default workload = BASELINE.json configs[1]: cPongDouble-v0, 65 536 envs per GPU, raw
(N, 2, 210, 160, 3) uint8 observations (1 env-step = 1 frame).  Prints ONE JSON line on rank 0.

Envs shard across GPUs with no data-path collective (weak scaling: 65 536 envs per GPU);
`--gather obs` adds the RCCL all-gather of BASELINE config #5 (xGMI-bound by construction).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic HBM bytes per env per launch, DESIGN.md "Kernels" (201 713 B/env-step
# for raw = 201 600 obs + 113 state/scalars; the frame-descriptor hand-off adds 8 B each way)
RAW_RASTER_BYTES = 2 * 100800 + 8
FUSED_RASTER_BYTES = {84: 2 * 4 * 84 * 84 + 8 * 8}
# CarRacing raster: obs 2*96*96 stored + ~1.6 KB body/joint state + ~4.7 KB track read (~24 900 B/env-step)
CAR_STEP_BYTES = 24900
HBM_PEAK = 8.0e12  # MI355X_MICROARCH.md: 8 TB/s spec


def cpu_baseline(workload, budget_s=12.0):
    """The oracle (CPU restatement of the reference path) timed on this host's cores on a
    bounded sample of the same workload.  Reported baseline, never the product path."""
    import numpy as np

    from competitive_rl_amd import _native
    from oracle import pong_oracle as po

    cores = len(os.sched_getaffinity(0))
    atlas = _native.load_score_atlas()
    if workload == "car":
        return cpu_baseline_car(cores, budget_s)
    if workload == "fused84_newest":
        workload = "fused84"
    if workload == "raw":
        n = 64 * cores
        env = po.PongOracle(n, atlas, obs_mode=po.RAW, seed=0)
        threads = cores
    else:
        n = 16 * cores
        env = po.PongOracle(n, atlas, obs_mode=po.GRAY, resized_dim=84, frame_stack=4, seed=0)
        threads = cores
    env.set_threads(threads)
    env.reset()
    rs = np.random.RandomState(0)
    acts = rs.randint(0, 3, (8, n, 2)).astype(np.int32)
    env.step(acts[0])
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < budget_s:
        env.step(acts[k % 8])
        k += 1
    dt = time.perf_counter() - t0
    env.close()
    return {"value": n * k / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"{k} steps x {n} envs, oracle/pong_oracle.c ({workload}), OpenMP over {threads} threads, {dt:.1f} s"}


def cpu_baseline_car(cores, budget_s):
    """Oracle CarRacing envs stepped + rendered serially on one core (scalar port)."""
    import ctypes as C

    import numpy as np

    from oracle import car_oracle as co

    L = co.lib()
    L.car_oracle_render.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    rs = np.random.RandomState(0)
    envs = []
    for i in range(4):
        e = co.CarEnv()
        while e.reset(rs.random_sample(24 * 8), i % 2) < 0:
            pass
        e.step(None)
        envs.append(e)
    out = np.zeros((96, 96), np.uint8)
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < budget_s:
        for e in envs:
            e.step(rs.uniform(-1, 1, (2, 2)))
            for v in range(2):
                L.car_oracle_render(e.buf.ctypes.data, v, out.ctypes.data)
        k += 1
    dt = time.perf_counter() - t0
    return {"value": len(envs) * k / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{k} steps x {len(envs)} envs, oracle/car_oracle.c step + 2 renders, 1 thread, {dt:.1f} s"}


POLICY_FLOP_PER_ENV = 2 * 516800  # LightActorCritic: conv1 409 600 + conv2 102 400 + actor 4 800 multiply-adds
FP32_VECTOR_PEAK = 157.3e12       # MI355X packed-fp32 vector peak (256 CUs x 4 SIMDs x 32 FMA lanes x 2.4 GHz x 2)


def cpu_baseline_tournament(budget_s=12.0):
    """CPU restatement of the same loop: oracle Pong env (42x42) + numpy LightActorCritic opponent."""
    import numpy as np

    from oracle import policy_oracle as P
    from oracle import pong_oracle as po

    atlas = np.load(os.path.join(ROOT, "competitive_rl_amd", "assets", "pong_score_atlas.npz"))["atlas"]
    cores = len(os.sched_getaffinity(0))
    n = 256
    env = po.PongOracle(n, atlas, obs_mode=po.GRAY, resized_dim=42, frame_stack=1, seed=0)
    env.set_threads(cores)
    pol = P.PolicyOracle(P.load_weights(os.path.join(ROOT, "competitive_rl_amd", "assets", "pong_policy_medium.npz")), n)
    obs = env.reset().copy()
    rs = np.random.RandomState(0)
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s:
        opp = pol(obs[:, 1]).reshape(-1)
        obs, _, _ = env.step(np.stack([rs.randint(0, 3, n), opp], 1))
        k += 1
    dt = time.perf_counter() - t0
    env.close()
    return {"value": n * k / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{k} steps x {n} envs, oracle/pong_oracle.c (42x42) + oracle/policy_oracle.py (numpy/BLAS), {dt:.1f} s"}


def bench_tournament(args, crl, torch, dist, dev, world, rank, n):
    """SURVEY 8f N2 + N4: cPongTournament-v0 (42x42) against the MEDIUM CNN opponent, everything on the device."""
    tour = crl.make_envs("cPongTournament-v0", num_envs=n, log_dir=None, seed=0, device=dev, env_id_base=rank * n)
    tour.reset_opponent("MEDIUM")
    tour.reset()
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    pool = [torch.randint(0, 3, (n,), generator=g, device=dev, dtype=torch.int32) for _ in range(16)]
    pol = tour.current_agent
    for i in range(args.warmup):
        tour.step_device(pool[i % 16])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    act = pol.act_device

    def timed_act(*a, **k):  # HIP events around the policy kernel, on the stream it is launched on (torch's current)
        e0, e1 = ev[timed_act.i]
        e0.record()
        r = act(*a, **k)
        e1.record()
        timed_act.i += 1
        return r

    timed_act.i = 0
    pol.act_device = timed_act
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        tour.step_device(pool[i % 16])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    k_us = sum(a.elapsed_time(b) for a, b in ev) / args.steps * 1e3
    tour.close()
    if rank == 0:
        achieved = POLICY_FLOP_PER_ENV * n / (k_us * 1e-6)
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic_tournament.json")
        if os.path.exists(tfile):
            traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
        line = {
            "metric": "env-steps/sec (whole node), cPongTournament 65536 envs per GPU vs the MEDIUM CNN opponent",
            "value": world * n * args.steps / dt, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cPongTournament-v0 {n} envs/GPU, 42x42 obs, opponent = reference checkpoint-medium (LightActorCritic) "
                                   "served by the HIP policy kernel, 1 step = 4 frames + 1 opponent forward pass (SURVEY 8f N2+N4)",
                       "envs_per_gpu": n, "actions": "uniform {0,1,2}, pre-generated on device", "auto_reset": True},
            "roofline": {"bound": "valu_fp32", "kernel": "pong_policy_light_kernel", "achieved": achieved / 1e12,
                         "peak": FP32_VECTOR_PEAK / 1e12, "unit": "TFLOP/s", "frac": achieved / FP32_VECTOR_PEAK, "traffic": traffic,
                         "flop_per_launch": POLICY_FLOP_PER_ENV * n, "avg_kernel_us": k_us, "launches_timed": args.steps,
                         "note": "fp32 vector FMAs (v_pk_fma_f32), not MFMA: the reference plays argmax of fp32 logits; "
                                 "a plain v_fma_f32 stream peaks at 78.6, packed fp32 measured at 134-142 TFLOP/s on this chip"},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_tournament()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=["raw", "fused84", "fused84_newest", "car", "tournament"], default="raw")
    ap.add_argument("--envs", type=int, default=None, help="envs per GPU (default 65536; 16384 for car)")
    ap.add_argument("--gather", choices=["none", "scalars", "obs"], default="none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import competitive_rl_amd as crl

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    n = args.envs or (16384 if args.workload == "car" else 65536)
    if args.workload == "tournament":
        return bench_tournament(args, crl, torch, dist, dev, world, rank, n)
    if args.workload == "car":
        env = crl.HipCarVecEnv(n, seed=0, device=dev, env_id_base=rank * n)
        raster_bytes, kernel = CAR_STEP_BYTES, "car_raster_kernel"
        desc = (f"cCarRacingDouble-v0 {n} envs/GPU, (N,2,96,96) u8 obs + Box2D-style car dynamics, 1 step = 1 CarRacing.step "
                "(BASELINE config #4)")
    elif args.workload == "raw":
        env = crl.HipPongVecEnv(n, seed=0, mode="raw", device=dev, env_id_base=rank * n)
        raster_bytes, kernel = RAW_RASTER_BYTES, "pong_raster_raw_sweep_kernel"
        desc = f"cPongDouble-v0 {n} envs/GPU raw (N,2,210,160,3) u8, 1 step = 1 frame (BASELINE config #2)"
    elif args.workload == "fused84_newest":
        # variant of config #3: only the newest plane is written, the consumer keeps the stack
        env = crl.HipPongVecEnv(n, seed=0, mode="wrapped", resized_dim=84, frame_stack=1, device=dev, env_id_base=rank * n)
        raster_bytes, kernel = 2 * 84 * 84 + 16, "pong_raster_gray_env_kernel"
        desc = (f"cPongDouble-v0 {n} envs/GPU fused skip4+max2+gray+84x84 INTER_AREA, newest plane only (N,2,1,84,84) u8, "
                "1 step = 4 frames (variant of BASELINE config #3)")
    else:
        env = crl.HipPongVecEnv(n, seed=0, mode="wrapped", resized_dim=84, frame_stack=4, device=dev,
                                env_id_base=rank * n)
        raster_bytes, kernel = FUSED_RASTER_BYTES[84], "pong_raster_gray_env_kernel"
        desc = (f"cPongDouble-v0 {n} envs/GPU fused skip4+max2+gray+84x84 INTER_AREA+4-stack (N,2,4,84,84) u8, "
                "1 step = 4 frames (BASELINE config #3)")
    env.reset()
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    if args.workload == "car":
        pool = [torch.rand((n, 2, 2), generator=g, device=dev, dtype=torch.float32) * 2 - 1 for _ in range(16)]
    else:
        pool = [torch.randint(0, 3, (n, 2), generator=g, device=dev, dtype=torch.int32) for _ in range(16)]

    def gather(buf, rew, done):
        if world == 1 or args.gather == "none":
            return
        if args.gather == "scalars":
            crl.all_gather_step((rew, done))
        else:
            crl.all_gather_step((buf, rew, done))

    for i in range(args.warmup):
        out = env.step_device(pool[i % 16])
        gather(*out)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    env.kernel_time_ms(0), env.kernel_time_ms(1)
    env.kernel_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = env.step_device(pool[i % 16])
        gather(*out)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    env.kernel_timing(False)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    dyn_ms, dyn_n = env.kernel_time_ms(0)
    ras_ms, ras_n = env.kernel_time_ms(1)
    done_frac = float(out[2].float().mean().item())
    env.close()

    if rank == 0:
        value = world * n * args.steps / dt
        ras_avg_s = ras_ms / max(ras_n, 1) * 1e-3
        achieved = raster_bytes * n / ras_avg_s if ras_avg_s > 0 else 0.0
        traffic = None
        tfile = os.path.join(ROOT, "profiles", f"traffic_{args.workload}.json")
        if os.path.exists(tfile):
            traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
        line = {
            "metric": ("env-steps/sec (whole node), cCarRacingDouble 16384 envs per GPU" if args.workload == "car"
                       else "env-steps/sec (whole node), cPongDouble 65536 envs per GPU"),
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": desc, "envs_per_gpu": n, "gather": args.gather if world > 1 else "n/a",
                       "actions": ("uniform [-1,1]^2 per car" if args.workload == "car" else "uniform {0,1,2}") + ", pre-generated on device",
                       "auto_reset": True},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": traffic,
                         "bytes_per_launch": raster_bytes * n, "avg_kernel_us": ras_avg_s * 1e6,
                         "launches_timed": ras_n,
                         "dynamics_kernel_avg_us": dyn_ms / max(dyn_n, 1) * 1e3},
        }
        if args.workload == "car":
            line["roofline"]["note"] = ("reported against HBM as required; the path is ALU/latency bound (DESIGN.md 4b). avg_kernel_us = the "
                                        "raster window of a step: launches for the three env classes incl. the wait for the side stream "
                                        "(coupled solve, resets); one full-batch launch alone takes ~1.66 ms")
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.workload)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
