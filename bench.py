#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the cPongDouble hot path on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload all|raw|fused84|fused84_f32|fused84_f32_ref|fused84_newest|car|car_fma|
                    tournament|tournament_full|protocol] [--envs E] [--gather none|scalars|obs|descriptors] [--no-cpu-baseline]

A "step" is one VecEnv.step over this rank's shard of envs with synthetic (pre-generated, device-resident) random
actions, auto-reset included, no host sync inside the timed loop.  The headline (``metric`` / ``value``) is
BASELINE.json configs[1]: cPongDouble-v0, 65 536 envs per GPU, raw (N, 2, 210, 160, 3) uint8 observations, 1 env-step =
1 frame.  With the default ``--workload all`` on one GPU the same process then measures every other single-GPU
configuration (fused gray+84x84+4-stack u8 and its float32 variant, cCarRacingDouble, the tournament loop) the same way
and attaches them under ``"configs"``, each with its own ``roofline`` and ``cpu_baseline``, plus a flat ``"configs_brief"``
{name: [env-steps/s, ms/step, roofline frac]}.  ONE JSON line on rank 0, numbers only (< 6 KB: the driver keeps 8 KB of stdout);
what the numbers mean -- workload definitions, roofline models, CPU baseline samples -- is DESIGN.md section 7.

``--gpus N`` without a launcher starts N fresh worker processes itself (one per GPU, before anything touches the GPU);
under ``torch.distributed.run`` the RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* of the environment are used.  Envs shard
across GPUs with no data-path collective (weak scaling: 65 536 envs per GPU); ``--gather obs|scalars`` adds the RCCL
all-gather of BASELINE config #5 (one packed collective per step on a side stream, overlapped with the next step).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# ---- algorithmic HBM bytes per env per launch of the dominant kernel (DESIGN.md "Kernels")
RAW_BYTES = 2 * 100800 + 8                 # both RGB views stored + the 8-byte frame descriptor read
FUSED_BYTES = 2 * 4 * 84 * 84 + 8 * 8      # (2, 4, 84, 84) u8 stored + the eight descriptors of the stack
FUSED_F32_BYTES = 4 * 2 * 4 * 84 * 84 + 8 * 8   # float32 output (both the widened and the reference's unrounded values)
NEWEST_BYTES = 2 * 84 * 84 + 16
CAR_BYTES = 24900                          # obs 18 432 + state r/w ~1 600 + track read ~4 700 + visited bits
HBM_PEAK = 8.0e12                          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s is what a float4 copy sustains)
FP32_PEAK = 157.3e12                       # fp32 vector = fp32 matrix peak (MI355X_MICROARCH.md)
POLICY_FLOP = 2 * 516800                   # LightActorCritic: conv1 409 600 + conv2 102 400 + actor 4 800 multiply-adds
POLICY_FULL_FLOP = 2 * (409600 + 991232 + 991232 + 768)  # ActorCritic: conv1 20x20x16x64, conv2 11x11x32x256, conv3 256x3872, actor 3x256
# cCarRacingDouble f32 FLOP per env-step, counted from the code paths (DESIGN.md 4b "FLOP model"): island solve
# 2 cars x (180 velocity iterations x 4 joints x 58 + joint init 4 x 70 + positions ~3 x 4 x 120 + integration 5 x 14)
# = 87 600; wheel model 2 x 4 x 64 (f64, counted once); sensor contacts ~2 x (300 AABB tests x 4 + 6 narrow pairs x 820)
# = 12 240; raster 2 views x 9 216 px x ~38 (inverse map 10, ~4 candidate tiles x 5 edges x ... resolved per 8x8 cell,
# car polygons, luma) = 700 400.  Total ~0.80 MFLOP; profiles/flops_car.json (SQ_INSTS_VALU_* counters) overrides it.
CAR_FLOP_MODEL = 87600 + 512 + 12240 + 700400


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="all",
                    choices=["all", "raw", "fused84", "fused84_f32", "fused84_f32_ref", "fused84_newest", "car", "car_fma", "tournament", "tournament_full", "protocol"])
    ap.add_argument("--envs", type=int, default=None, help="envs per GPU (default 65536; 16384 for car)")
    ap.add_argument("--gather", choices=["none", "scalars", "obs", "descriptors"], default="none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def spawn_ranks(args):
    """``python bench.py --gpus N`` with no launcher around it: N fresh children, one per GPU.  Runs before torch is
    imported -- a process that has touched the GPU is never re-executed; the parent only waits."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = [p.wait() for p in procs]
    sys.exit(max(abs(rc) for rc in rcs))


# =============================================================================================== CPU baselines
def effective_cores():
    """The CPUs this process may actually keep busy: its affinity mask, capped by the cgroup's CPU quota.  (Round 6: the GPU boxes show 256
    hardware threads in the affinity mask under a quota of 16 CPUs -- `cpu.max` = 1600000 100000 --, so rounds 1-5's 256 workers / 256 OpenMP
    threads were throttled 16-fold and time-sliced: the same C code does 49 k env-steps/s on 16 threads and 14 k on 256, tools/cpu_legs.py.)"""
    visible = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if q != "max":
            quota = int(q) / int(period)
    except (OSError, ValueError):
        try:  # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    cores = visible if quota is None else max(1, min(visible, int(quota + 0.5)))
    return cores, visible, quota


def cpu_baselines(workloads, budget_s=float(os.environ.get("CRL_BENCH_CPU_BUDGET_S", "8.0"))):
    """Rank 0, one GPU only, BEFORE the GPU is touched: the oracle timed on this host's cores on bounded samples of
    the same workloads -- (a) in the reference's SubprocVecEnv architecture (one process per env, pipes, pickled
    observations; oracle/subproc_baseline.py), which is the reported ``value``; (b) as one OpenMP batch over all
    cores (upper bound of a compiled port); plus BASELINE config #1 (4 envs, synchronous).  ``cores`` = effective_cores(): the
    affinity mask capped by the cgroup's CPU quota -- one worker / one thread per CPU the process can really use.  Never the product path."""
    import numpy as np

    from oracle import pong_oracle as po
    from oracle import subproc_baseline as sb

    cores, visible, quota = effective_cores()
    host = {"host_threads_visible": visible, "cpu_quota": quota}
    atlas = np.load(os.path.join(ROOT, "competitive_rl_amd", "assets", "pong_score_atlas.npz"))["atlas"]
    out = {}

    def openmp_port(kind, n):
        env = po.PongOracle(n, atlas, obs_mode=po.RAW if kind == "raw" else po.GRAY, resized_dim=84 if kind == "fused84" else 42,
                            frame_stack=4 if kind == "fused84" else 1, seed=0)
        env.set_threads(cores)
        env.reset()
        rs = np.random.RandomState(0)
        acts = rs.randint(0, 3, (8, n, 2)).astype(np.int32)
        env.step(acts[0])
        t0, k = time.perf_counter(), 0
        while time.perf_counter() - t0 < budget_s * 0.6:
            env.step(acts[k % 8])
            k += 1
        dt = time.perf_counter() - t0
        env.close()
        return {"value": n * k / dt, "unit": "env-steps/s", "cores": cores, "kind": "port", "sample": f"{k} steps x {n} envs, one OpenMP batch, {dt:.1f} s"}

    def subproc(kind, label):
        v, k, dt = sb.time_subproc(kind, cores, budget_s)
        return {"value": v, "unit": "env-steps/s", "cores": cores, "kind": "port", "architecture": "subproc",
                "sample": f"{k} lock-steps x {cores} one-env worker processes ({label}), {dt:.1f} s"}

    cache = {}

    def once(key, fn):  # fused84 / fused84_f32 / fused84_f32_ref share one CPU workload, car / car_fma another
        if key not in cache:
            cache[key] = fn()
        return cache[key]

    for wl in workloads:
        if wl == "raw":
            b = dict(once("raw", lambda: subproc("raw", "raw 2x(210,160,3) u8")))
            b["openmp_port"] = once("raw_omp", lambda: openmp_port("raw", 64 * cores))
        elif wl in ("fused84", "fused84_f32", "fused84_f32_ref", "fused84_newest", "protocol"):
            b = dict(once("gray", lambda: subproc("gray_84", "skip4+max2+gray+84x84")))
            b["openmp_port"] = once("gray_omp", lambda: openmp_port("fused84", 16 * cores))
        elif wl in ("car", "car_fma"):
            b = once("car", lambda: subproc("car", "Box2D-style step + two 96x96 renders"))
        elif wl == "tournament":
            b = cpu_baseline_tournament(po, atlas, cores, budget_s * 0.8)
        elif wl == "tournament_full":
            b = cpu_baseline_tournament(po, atlas, cores, budget_s * 0.8, full=True)
        else:
            continue
        out[wl] = dict(b, **host)
    if "raw" in workloads:  # BASELINE config #1: make_envs(num_envs=4, asynchronous=False), 42x42, 1000 steps
        v, k, dt = sb.time_dummy("gray_42", 4, 1000)
        out["raw"]["config1_dummy_n4"] = {"value": v, "unit": "env-steps/s", "cores": 1, "kind": "port", "sample": f"{k} steps x 4 envs, one process, {dt:.1f} s"}
    return out


def full_policy_weights(seed=0):
    """Random-init ActorCritic tensors (He-normal weights, small biases) for the tournament_full workload: the reference tree
    ships no checkpoint of that network, and the forward pass costs the same whatever the values."""
    import numpy as np

    rs = np.random.RandomState(seed)
    shapes = {"conv1": (16, 4, 4, 4), "conv2": (32, 16, 4, 4), "conv3": (256, 32, 11, 11), "actor": (3, 256), "critic": (1, 256)}
    w = {}
    for k, shp in shapes.items():
        w[k + "_w"] = (rs.standard_normal(shp) * np.sqrt(2.0 / np.prod(shp[1:]))).astype(np.float32)
        w[k + "_b"] = (rs.standard_normal(shp[0]) * 0.1).astype(np.float32)
    return w


def cpu_baseline_tournament(po, atlas, cores, budget_s, full=False):
    """CPU restatement of the tournament loop: oracle Pong env (42x42) + numpy LightActorCritic / ActorCritic opponent."""
    import numpy as np

    from oracle import policy_oracle as P

    n = 256
    env = po.PongOracle(n, atlas, obs_mode=po.GRAY, resized_dim=42, frame_stack=1, seed=0)
    env.set_threads(cores)
    if full:
        pol = P.PolicyOracle(full_policy_weights(), n, full=True)
    else:
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
    return {"value": n * k / dt, "unit": "env-steps/s", "cores": cores, "kind": "port", "sample": f"{k} steps x {n} envs, C env (OpenMP) + numpy policy, {dt:.1f} s"}


# =============================================================================================== GPU workloads
def traffic_of(name):
    """HBM bytes per launch of the dominant kernel from the PMC passes of tools/profile_gpu.sh (profiles/traffic_*.json):
    measured in a separate profiled run of this same command, not in this process."""
    f = os.path.join(ROOT, "profiles", f"traffic_{name}.json")
    if not os.path.exists(f):
        return None
    return json.load(open(f)).get("hbm_bytes_per_launch")


# one short line per workload (what each one is, in full: DESIGN.md section 7)
WORKLOADS = {
    "raw": "cPongDouble-v0 raw (N,2,210,160,3) u8, 1 step = 1 frame (BASELINE config #2)",
    "fused84": "cPongDouble-v0 fused skip4+max2+gray+84x84+4-stack (N,2,4,84,84) u8, 1 step = 4 frames (config #3)",
    "fused84_f32": "config #3 with float32 output: the uint8 values widened",
    "fused84_f32_ref": "config #3 with float32 output: the reference's own unrounded step() values (obs_dtype=float32_ref)",
    "fused84_newest": "config #3, newest plane only (N,2,1,84,84) u8",
    "car": "cCarRacingDouble-v0 (N,2,96,96) u8 + Box2D-style dynamics with car-car contacts (config #4), steady state",
    "car_fma": "config #4, island solver iterations in fused multiply-adds (CRL_FLAG_CAR_FMA), steady state",
    "tournament": "cPongTournament-v0 42x42, MEDIUM LightActorCritic opponent on the HIP policy kernel",
    "tournament_full": "cPongTournament-v0 42x42, full-size ActorCritic opponent (random-init weights) on the HIP MFMA kernels",
    "protocol": "cPongDouble-v0 84x84 through the drop-in protocol: make_envs().step(), then step_envs + FrameStackTensor(4)",
}


def _r(x, nd=5):
    """numbers of the JSON line: 5 significant digits"""
    if isinstance(x, float):
        return float(f"{x:.{nd}g}")
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


def timed_loop(G, args, step, after=None):
    """W warm-up calls of step(i), then exactly K timed ones between barrier + synchronize on both sides; max over ranks."""
    torch, dist, dev, world = G["torch"], G["dist"], G["dev"], G["world"]
    dist_on = G["dist_on"]
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    if after:
        after()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    return t0


def run_protocol(args, G):
    """The reference-shaped calls around the same env: (a) step_device, the library's hot-loop entry;
    (b) make_envs(...).step(device actions) -- VecEnv.step_async / step_wait with cloned outputs and lazy infos; (c) the same with
    infos[i] read for the envs that finished (one host copy of the done flags per step, the terminal observations drawn by one call);
    (d) step_envs + FrameStackTensor (reference utils/utils.py:23-60, 145-173): the trainer's books, observation stack of agent 0 as
    float32 (N, 4, 84, 84) kept on the device -- since round 6 BOUND to the env: envs.step draws the stack's next state in the launch
    that draws the observation (crl_step_stack) and update() is a pointer swap; (e) the same with an opt-in uint8 stack; (f) round 5's
    path for comparison: the stack rolled and appended by a kernel of its own (13.4 GB per step).
    Every leg: W warm-up calls, K timed ones between barrier + synchronize on both sides, max over ranks."""
    torch, crl, dist, dev, rank, dist_on = G["torch"], G["crl"], G["dist"], G["dev"], G["rank"], G["dist_on"]
    import numpy as np

    n = args.envs or 65536
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    pool = [torch.randint(0, 3, (n, 2), generator=gen, device=dev, dtype=torch.int32) for _ in range(16)]
    env = crl.make_envs("cPongDouble-v0", num_envs=n, log_dir=None, seed=0, resized_dim=84, frame_stack=None, device=dev, env_id_base=rank * n)
    books = dict(ep=torch.zeros((n, 2), dtype=torch.float32, device=dev), rr=[], lr=[], steps=0, episodes=0)
    sink = [0]
    cur = {}

    def leg_device(i):
        env.step_device(pool[i % 16])

    def leg_step(i):
        env.step(pool[i % 16])

    def leg_infos(i):
        obs, rew, done, infos = env.step(pool[i % 16])
        for j in np.flatnonzero(done[:, 0].cpu().numpy()):
            sink[0] += len(infos[int(j)]["terminal_observation"])

    def leg_step_envs(i):
        out = crl.step_envs(pool[i % 16], env, books["ep"], cur["fst"], books["rr"], books["lr"], books["steps"], books["episodes"], dev, False)
        books["episodes"], books["steps"] = out[5], out[6]

    def stack(dtype, bind):
        """the training scripts' preamble: a fresh stack, the env reset, its first observation pushed"""
        if cur.get("fst") is not None:
            cur["fst"].unbind()
        cur["fst"] = None
        torch.cuda.empty_cache()
        f = crl.FrameStackTensor(n, (1, 84, 84), 4, dev, dtype=dtype)
        f._bind_tried = not bind  # (False: step_envs binds it on its first call)
        f.update(env.reset()[0])
        cur["fst"] = f

    env.reset()
    res, fused = {}, {}
    legs = (("step_device", leg_device, None), ("vec_env_step", leg_step, None), ("vec_env_step_infos", leg_infos, None),
            ("step_envs", leg_step_envs, (torch.float32, True)), ("step_envs_u8_stack", leg_step_envs, (torch.uint8, True)),
            ("step_envs_unbound", leg_step_envs, (torch.float32, False)))
    kernel_us = {}
    for key, leg, st in legs:
        if st is not None:
            stack(*st)
        timed = key in ("step_envs", "step_envs_u8_stack") and not os.environ.get("CRL_BENCH_NO_KERNEL_TIMING")

        def arm():  # hipEvents around the draw of the timed steps only (the warm-up's launches are not counted)
            if timed:
                env.kernel_time_ms(1)
                env.kernel_timing(True)

        t0 = timed_loop(G, args, leg, after=arm)
        torch.cuda.synchronize()
        if timed:
            env.kernel_timing(False)
            ms, cnt = env.kernel_time_ms(1)
            kernel_us[key] = ms / max(cnt, 1) * 1e3
        if dist_on:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist_on:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        res[key] = dt / args.steps * 1e3
        if st is not None:
            fused[key] = cur["fst"].fused_updates
    env.close()
    base = res["step_device"]
    stack_bytes = 4 * 84 * 84 * 4
    out = {"value": G["world"] * n * args.steps / (res["step_envs"] * 1e-3 * args.steps), "unit": "env-steps/s", "ms_per_step": res["step_envs"], "dtype": "u8",
           "config": {"workload": WORKLOADS["protocol"], "envs_per_gpu": n, "stack": "float32 (N,4,84,84), bound to the env (drawn by the step)"},
           "legs_ms_per_step": res, "overhead_us_per_step": {k: (v - base) * 1e3 for k, v in res.items() if k != "step_device"},
           "fused_updates": fused, "episodes_recorded": books["episodes"]}
    if rank == 0 and kernel_us.get("step_envs"):
        # the launch that draws the float32 stack AND the uint8 observation: 4 x 28 224 + 2 x 7 056 B per env, by hipEvents on its stream;
        # leg_frac: the stack alone (112 896 B/env written once) over the WHOLE step_envs leg
        per_env = stack_bytes + 2 * 84 * 84
        ach = per_env * n / (kernel_us["step_envs"] * 1e-6)
        out["roofline"] = {"bound": "hbm", "kernel": "pong_raster_gray_env_kernel<STACK>", "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                           "frac": ach / HBM_PEAK, "traffic": traffic_of("protocol"), "bytes_per_launch": per_env * n, "avg_kernel_us": kernel_us["step_envs"],
                           "leg_frac": stack_bytes * n / (res["step_envs"] * 1e-3) / HBM_PEAK,
                           "u8_stack_kernel_us": kernel_us.get("step_envs_u8_stack")}
    return out


def run_workload(name, args, G):
    """Builds the env for `name`, does W warm-up steps, times exactly K steps (barrier + synchronize on both sides, max
    over ranks) and returns the result dict (value, ms_per_step, roofline ...)."""
    if name == "protocol":
        return run_protocol(args, G)
    torch, crl, dist, dev, world, rank = G["torch"], G["crl"], G["dist"], G["dev"], G["world"], G["rank"]
    is_car = name in ("car", "car_fma")
    n = args.envs or (16384 if is_car else 65536)
    base = rank * n
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    policy_events = None
    if name in ("tournament", "tournament_full"):
        env = crl.make_envs("cPongTournament-v0", num_envs=n, log_dir=None, seed=0, device=dev, env_id_base=base)
        if name == "tournament_full":
            env.add_agent("FULL", crl.Policy(crl.tournament.single_obs_space, crl.tournament.single_act_space, n, use_light_model=False, device=dev,
                                             weights=full_policy_weights()))
            env.reset_opponent("FULL")
        else:
            env.reset_opponent("MEDIUM")
        pool = [torch.randint(0, 3, (n,), generator=gen, device=dev, dtype=torch.int32) for _ in range(16)]
        pol, act = env.current_agent, env.current_agent.act_device
        policy_events = []
        event_pool = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]  # made before the timed loop

        def timed_act(*a, **k):  # HIP events around the policy kernel, on the stream it is launched on (torch's current)
            if timed_act.on and len(policy_events) < len(event_pool):
                e0, e1 = event_pool[len(policy_events)]
                e0.record()
                r = act(*a, **k)
                e1.record()
                policy_events.append((e0, e1))
                return r
            return act(*a, **k)

        timed_act.on = False
        pol.act_device = timed_act
        inner = env.env
        kernel, dtype = "pong_policy_mfma_kernel", "f32"
    elif is_car:
        env = inner = crl.HipCarVecEnv(n, seed=0, device=dev, env_id_base=base, solver="fma" if name == "car_fma" else "box2d")
        pool = [torch.rand((n, 2, 2), generator=gen, device=dev, dtype=torch.float32) * 2 - 1 for _ in range(16)]
        kernel, dtype = "car_obs_third_kernel", "f32"
    else:
        kw = {"raw": dict(mode="raw"),
              "fused84": dict(mode="wrapped", resized_dim=84, frame_stack=4),
              "fused84_f32": dict(mode="wrapped", resized_dim=84, frame_stack=4, obs_dtype="float32"),
              "fused84_f32_ref": dict(mode="wrapped", resized_dim=84, frame_stack=4, obs_dtype="float32_ref"),
              "fused84_newest": dict(mode="wrapped", resized_dim=84, frame_stack=1)}[name]
        env = inner = crl.HipPongVecEnv(n, seed=0, device=dev, env_id_base=base, **kw)
        pool = [torch.randint(0, 3, (n, 2), generator=gen, device=dev, dtype=torch.int32) for _ in range(16)]
        kernel = "pong_raster_raw_sweep_kernel" if name == "raw" else "pong_gray_f32ref_kernel" if name == "fused84_f32_ref" else "pong_raster_gray_env_kernel"
        dtype = "f32" if name in ("fused84_f32", "fused84_f32_ref") else "u8"
    env.reset()
    episodes_before = None
    if is_car:
        # Steady state, un-timed: gym's TimeLimit counters staggered uniformly over [0, 1000) and ONE episode length of pre-roll, so
        # that every env has been reset once at a different time -- the batch then holds cars of every age (spread over their
        # tracks, ~10 % of the envs with overlapping cars), and every timed step carries its ~n/1000 resets (terminal frames,
        # reset, map raster, walk-ahead of the next track).  A run timed right after a synchronous reset has none of that.
        st = env.get_state()
        st["elapsed"] = (torch.arange(n, dtype=torch.int64) * 1000 // n).numpy().astype(st["elapsed"].dtype)
        env.set_state(st)
        for i in range(int(os.environ.get("CRL_BENCH_CAR_PREROLL", "1000"))):
            env.step_device(pool[i % 16])
        torch.cuda.synchronize()

    gather_state, dist_on = G.get("gather"), G["dist_on"]

    def step(i):
        if dist_on and args.gather == "obs" and name not in ("tournament", "tournament_full"):
            # the env draws straight into the collective's send buffer (two of them, alternating: gather(t) still reads one)
            out = env.step_device(pool[i % 16], obs_out=gather_state.obs_slot(inner._obs[0].shape, inner._obs[0].dtype, dev))
        else:
            out = env.step_device(pool[i % 16])
        if dist_on and args.gather != "none":
            if args.gather == "descriptors":
                # 64 bytes of frame descriptors per env over the links; every rank then re-draws the GLOBAL batch (inside wait())
                gather_state.wait(materialize=True)
                gather_state.launch(out[1:], env=inner)
            else:
                # at most one step's collectives in flight: gather(t) overlaps simulate(t+1).  materialize=True: the step ends holding
                # the usable global tensors -- the observation is the collective's receive buffer itself (sharding.StepGather), so this
                # costs the slicing of the few-bytes-per-env fields only
                G["gathered"] = gather_state.wait(materialize=True)
                gather_state.launch(out if args.gather == "obs" else out[1:])

    warm_resets = None
    if is_car:
        # the episode counters BEFORE the warm-up (reading the state takes the host a while; between the warm-up and the timed
        # region it would leave the GPU idle long enough for its first kernels to start 10-20 ms late -- measured: a 20-step window
        # then reads 1.3-2.1 ms per step instead of 0.9); the warm-up's own resets are counted on the device and subtracted
        episodes_before = env.get_state()["episode"].astype("int64").sum()
        warm_resets = torch.zeros((), dtype=torch.int64, device=dev)

    for i in range(args.warmup):
        step(i)
        if warm_resets is not None:
            warm_resets += inner._done.sum()  # (the env's device-side done flags of this step; same stream)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    inner.kernel_time_ms(0), inner.kernel_time_ms(1)
    inner.kernel_timing(not os.environ.get("CRL_BENCH_NO_KERNEL_TIMING"))  # (A/B: what the hipEvent brackets themselves cost)
    if policy_events is not None:
        timed_act.on = True
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    if dist_on and args.gather != "none":
        G["gathered"] = gather_state.wait(materialize=True)
    if os.environ.get("CRL_BENCH_DEBUG"):  # (where a short window's time goes: the host's enqueueing, the caller's stream, the rest of the device)
        t_host = time.perf_counter() - t0
        torch.cuda.current_stream().synchronize()
        t_stream = time.perf_counter() - t0
        torch.cuda.synchronize()
        print(f"[{name}] host enqueue {t_host * 1e3:.2f} ms, caller's stream done {t_stream * 1e3:.2f} ms, device done {(time.perf_counter() - t0) * 1e3:.2f} ms",
              file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    dt = time.perf_counter() - t0
    inner.kernel_timing(False)
    if dist_on:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    dyn_ms, dyn_n = inner.kernel_time_ms(0)
    ras_ms, ras_n = inner.kernel_time_ms(1)
    touch_ms, touch_n, touch_max = inner.kernel_time_stats(2) if is_car else (0.0, 0, 0.0)
    final_state = env.get_state() if episodes_before is not None else None
    resets = int(final_state["episode"].astype("int64").sum() - episodes_before - int(warm_resets.item())) if episodes_before is not None else None
    env.close()
    res = {"value": world * n * args.steps / dt, "unit": "env-steps/s", "ms_per_step": dt / args.steps * 1e3, "dtype": dtype,
           "config": {"workload": WORKLOADS[name], "envs_per_gpu": n}}
    if resets is not None:
        res["config"]["resets_in_timed_region"] = resets
    if rank != 0:
        return res
    if name == "tournament_full":
        k_us = sum(a.elapsed_time(b) for a, b in policy_events) / max(len(policy_events), 1) * 1e3
        ach = POLICY_FULL_FLOP * n / (k_us * 1e-6)
        res["roofline"] = {"bound": "mfma", "kernel": "policy_full_front+conv3+actor", "achieved": ach / 1e12, "peak": FP32_PEAK / 1e12, "unit": "TFLOP/s",
                           "frac": ach / FP32_PEAK, "traffic": None, "avg_kernel_us": k_us, "launches_timed": len(policy_events)}
        return res
    if name == "tournament":
        if os.environ.get("CRL_LIB_VARIANT") and os.environ.get("CRL_POLICY_MFMA", "3") != "3":
            return res  # (profiling build with another form of the network selected -- 0 packed FMA, 1 conv1 on the fp32 matrix instruction: the bf16 x 3 instruction mix priced below is not what ran)
        k_us = sum(a.elapsed_time(b) for a, b in policy_events) / max(len(policy_events), 1) * 1e3
        # matrix-pipe kernel: per tile of 16 conv2 positions (6.25 tiles per env) conv1 = 24 v_mfma_f32_16x16x32_bf16 (three exact
        # bf16 products per tap, 16 cycles each), conv2 = 16 v_mfma_f32_16x16x4_f32 (32 cycles): EXECUTED matrix FLOP against the issue-rate
        # peak of that instruction mix, i.e. frac = matrix-pipe time / kernel time; fp32_equivalent = the network's own 1.03 MFLOP per env
        tiles = n * 100 / 16
        exec_flop = tiles * (24 * 16 * 16 * 32 * 2 + 16 * 2048)
        floor_s = tiles * (24 * 16 + 16 * 32) / (1024 * 2.4e9)  # 1 024 SIMDs, 2.4 GHz (the clock the launch runs at)
        ach, peak = exec_flop / (k_us * 1e-6), exec_flop / floor_s
        res["roofline"] = {"bound": "mfma", "kernel": kernel, "achieved": ach / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": ach / peak,
                           "traffic": traffic_of(name), "avg_kernel_us": k_us, "launches_timed": len(policy_events),
                           "fp32_equivalent_tflops": POLICY_FLOP * n / (k_us * 1e-6) / 1e12}
        return res
    ras_s = ras_ms / max(ras_n, 1) * 1e-3
    bytes_per_env = {"raw": RAW_BYTES, "fused84": FUSED_BYTES, "fused84_f32": FUSED_F32_BYTES, "fused84_f32_ref": FUSED_F32_BYTES,
                     "fused84_newest": NEWEST_BYTES, "car": CAR_BYTES, "car_fma": CAR_BYTES}[name]
    drawn = 1.0
    if is_car:
        # the timed launch draws the envs that are neither coupled nor finished (the others' wavefronts exit at once; their frames come
        # from the list-driven launches): count only what it draws.  coupled: the last step's flags; finished: resets per step.
        drawn = 1.0 - float((final_state["coupled"] != 0).mean()) - resets / max(args.steps, 1) / n
    ach = bytes_per_env * n * drawn / ras_s if ras_s > 0 else 0.0
    res["roofline"] = {"bound": "hbm", "kernel": kernel, "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ach / HBM_PEAK,
                       "traffic": traffic_of(name) or (traffic_of("car") if is_car else None), "bytes_per_launch": bytes_per_env * n * drawn, "avg_kernel_us": ras_s * 1e6,
                       "launches_timed": ras_n, "dynamics_kernel_avg_us": dyn_ms / max(dyn_n, 1) * 1e3}
    if is_car:
        flop = CAR_FLOP_MODEL
        ff = os.path.join(ROOT, "profiles", "flops_car.json")
        if os.path.exists(ff):
            flop = json.load(open(ff))["f32_flop_per_env_step"]
        fl = flop * res["value"] / world
        res["roofline"]["envs_drawn_by_timed_launch"] = drawn
        # what bounds the STEP: SURVEY 8(d)'s formula -- env-steps/s x 24 900 B over the HBM peak -- and the kernel the step's
        # period follows: the touching solve, lone wavefronts on the caller's stream (hipEvents on that stream, crl_kernel_time_stats slot 2)
        res["roofline"]["step_frac"] = CAR_BYTES * res["value"] / world / HBM_PEAK
        res["roofline"]["critical_path"] = {"kernel": "car_touch_kernel", "bound": "dependent-issue latency of lone wavefronts (Gauss-Seidel islands)",
                                            "mean_us": touch_ms / max(touch_n, 1) * 1e3, "max_us": touch_max * 1e3, "launches_timed": touch_n,
                                            "step_us": dt / args.steps * 1e6}
        res["roofline_valu"] = {"bound": "valu_fp32", "achieved": fl / 1e12, "peak": FP32_PEAK / 1e12, "unit": "TFLOP/s", "frac": fl / FP32_PEAK,
                                "flop_per_env_step": flop}
    return res


ALL = ["raw", "fused84", "fused84_newest", "fused84_f32", "fused84_f32_ref", "car", "car_fma", "tournament", "tournament_full", "protocol"]


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        spawn_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    multi = args.workload == "all" and world == 1
    names = ALL if multi else [("raw" if args.workload == "all" else args.workload)]
    # the CPU baselines run first: worker processes are started while this process has not initialised the GPU
    cpu = {}
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baselines(names)

    import torch
    import torch.distributed as dist

    import competitive_rl_amd as crl

    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # CRL_BENCH_FORCE_DIST=1: a single rank goes through the process group all the same (RCCL world of one: barriers, the max over ranks, the
    # --gather collective) -- what a one-GPU box can run of the N > 1 path, tests/test_hip_round2.py
    dist_on = world > 1 or os.environ.get("CRL_BENCH_FORCE_DIST") == "1"
    G = dict(torch=torch, crl=crl, dist=dist, dev=dev, world=world, rank=rank, dist_on=dist_on)
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
        G["gather"] = crl.StepGather(overlap=True, mode="descriptors" if args.gather == "descriptors" else "obs")
    results = {}
    for nm in names:
        if os.environ.get("CRL_BENCH_STREAM"):  # (A/B: the loop on a created stream instead of the legacy default stream)
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                results[nm] = run_workload(nm, args, G)
        else:
            results[nm] = run_workload(nm, args, G)
        torch.cuda.empty_cache()
    if rank == 0:
        head = results[names[0]]
        metric = {"car": "env-steps/sec (whole node), cCarRacingDouble 16384 envs per GPU",
                  "car_fma": "env-steps/sec (whole node), cCarRacingDouble 16384 envs per GPU, island solver in fused multiply-adds",
                  "tournament": "env-steps/sec (whole node), cPongTournament 65536 envs per GPU vs the MEDIUM CNN opponent",
                  "tournament_full": "env-steps/sec (whole node), cPongTournament 65536 envs per GPU vs a full-size ActorCritic opponent",
                  "protocol": "env-steps/sec (whole node), cPongDouble 65536 envs per GPU through make_envs().step + step_envs"}.get(
            names[0], "env-steps/sec (whole node), cPongDouble 65536 envs per GPU")
        line = {"metric": metric, "value": head["value"], "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": head["dtype"],
                "data": "synthetic", "config": dict(head["config"], gather=args.gather if dist_on else "n/a", actions="uniform random, pre-generated on device",
                                                    auto_reset=True),
                "roofline": head.get("roofline")}
        for k in ("roofline_valu", "legs_ms_per_step", "overhead_us_per_step"):
            if k in head:
                line[k] = head[k]
        if names[0] in cpu:
            line["cpu_baseline"] = cpu[names[0]]
        if multi:
            # every configuration where a parser that keeps only top-level scalars / short lists still finds it: [env-steps/s, ms/step, roofline frac]
            line["configs_brief"] = {nm: [results[nm]["value"], results[nm]["ms_per_step"], (results[nm].get("roofline") or {}).get("frac")] for nm in names}
        # what ran where: one process per GPU, contiguous env ranges by global id, no data-path collective unless --gather asks for the
        # config-#5 all-gather; the collective library the ranks would use
        npg = head["config"]["envs_per_gpu"]
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as exc:  # (a build without the collective library)
            rccl = f"unavailable ({type(exc).__name__})"
        line["comm"] = {"backend": "nccl (= RCCL on ROCm)" if dist_on else "none (one rank)", "rccl_version": rccl, "world": world,
                        "env_ranges": [[r * npg, (r + 1) * npg] for r in range(world)], "gather": args.gather,
                        # every timed step ends with the global tensors in hand; "obs": the observation is the collective's receive buffer (no copy)
                        "materialized": bool(dist_on and args.gather != "none"),
                        "gathered_obs_shape": list(G["gathered"][0].shape) if G.get("gathered") else None}
        if multi:
            line["configs"] = {}
            for nm in names[1:]:  # slim: the name is the key of WORKLOADS / DESIGN.md section 7, numbers only
                r = {k: v for k, v in results[nm].items() if k != "unit"}
                r["config"] = {k: v for k, v in r["config"].items() if k != "workload"}
                r.pop("overhead_us_per_step", None)  # (= legs_ms_per_step minus its first entry)
                r.pop("episodes_recorded", None)
                if r.get("roofline"):
                    r["roofline"] = {k: v for k, v in r["roofline"].items() if k not in ("bytes_per_launch", "launches_timed", "dynamics_kernel_avg_us")}
                    if "critical_path" in r["roofline"]:  # kernel, mean / max / step in us (what bounds it: DESIGN.md 4.4)
                        r["roofline"]["critical_path"] = {k: v for k, v in r["roofline"]["critical_path"].items() if k in ("kernel", "mean_us", "max_us", "step_us")}
                if nm in cpu:
                    b = {k: v for k, v in cpu[nm].items() if k in ("value", "cores", "kind")}  # (unit: env-steps/s; the sample is described in DESIGN.md 7)
                    if "openmp_port" in cpu[nm]:
                        b["openmp_port_value"] = cpu[nm]["openmp_port"]["value"]
                    r["cpu_baseline"] = b
                line["configs"][nm] = r
        print(json.dumps(_r(line), separators=(",", ":")), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
