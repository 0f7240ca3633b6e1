"""Builds libcrl_hip.so (the C-ABI library of include/crl.h) in-tree with hipcc for gfx950.

    python -m competitive_rl_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off is required for parity: the game
arithmetic is f64/f32 with one rounding per operation (no FMA contraction).
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libcrl_hip.so")
SOURCES = ["crl_api.hip", "pong_dynamics.hip", "pong_raster_raw.hip", "pong_raster_gray.hip",
           "crl_car_api.hip", "car_step.hip", "car_contact.hip", "car_track.hip", "car_raster.hip", "car_obs.hip", "pong_policy.hip", "pong_policy_full.hip", "frame_stack.hip", "crl_selftest.hip"]
HEADERS = [os.path.join(CSRC, h) for h in ("pong_device.h", "car_device.h", "car_solver.h", "car_obs_tile.h", "crl_internal.h", "pong_policy_full.h", "pong_gray_tile.inc")] + [os.path.join(ROOT, "include", "crl.h"), os.path.join(ROOT, "include", "crl_rot.h"), os.path.join(ROOT, "include", "crl_f64.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fgpu-rdc" if False else "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for s in SOURCES:
        o = os.path.join(CSRC, s.replace(".hip", ".o"))
        extra = os.environ.get("CRL_EXTRA_FLAGS_" + s.split(".")[0].upper(), "").split()  # per-file experiments
        cmd = [hipcc, *FLAGS, *extra, "-I", os.path.join(ROOT, "include"), "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(o)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def build_variant(tag, flags, verbose=False, force=True):
    """A profiling build beside the shipped one: every source compiled with `flags` added, objects csrc/<tag>_*.o, library
    libcrl_hip_<tag>.so (loaded when CRL_LIB_VARIANT=<tag>)."""
    out = os.path.join(PKG, "libcrl_hip_%s.so" % tag)
    if not force and os.path.exists(out):
        t = os.path.getmtime(out)
        if not any(os.path.getmtime(d) > t for d in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS):
            return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for s in SOURCES:
        o = os.path.join(CSRC, tag + "_" + s.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, *flags, "-I", os.path.join(ROOT, "include"), "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(o)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs])
    return out


def build_abl(verbose=False):
    """The profiling variant (-DCRL_ABLATION): the superseded kernels (round 2's analytic CarRacing raster, the first Pong
    writers, the packed-FMA opponent network), the timing ablations that give WRONG output and the phase cycle stamps live
    only here; tests that A/B a superseded kernel against the checker load it with CRL_LIB_VARIANT=abl."""
    return build_variant("abl", ["-DCRL_ABLATION"], verbose=verbose, force=False)


def build_c_demo(verbose=False):
    """examples/c_abi_demo: a caller of include/crl.h without Python or torch (links libcrl_hip.so)."""
    build()
    src = os.path.join(ROOT, "examples", "c_abi_demo.cpp")
    out = os.path.join(ROOT, "examples", "c_abi_demo")
    if os.path.exists(out) and os.path.getmtime(out) > max(os.path.getmtime(src), os.path.getmtime(LIB)):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O2", "-I", os.path.join(ROOT, "include"), src, "-L", os.path.dirname(LIB), "-lcrl_hip",
           "-Wl,-rpath,$ORIGIN/../competitive_rl_amd", "-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    if "--variant" in sys.argv:  # python -m competitive_rl_amd.build --variant abl -DCRL_ABLATION
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:], verbose=True))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_c_demo(verbose=True))
