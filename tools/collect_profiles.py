#!/usr/bin/env python3
"""Copy the judged summaries of tools/profile_gpu.sh runs from gpurun_out/ (scratch) into
profiles/ (tracked) and derive profiles/traffic_<workload>.json (HBM bytes per launch of the
dominant kernel: WRITE_SIZE + 2 x FETCH_SIZE in KiB, MI355X_MICROARCH.md "HBM").

    python tools/collect_profiles.py r02 raw fused84 fused84_f32 car tournament
"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOMINANT = {"raw": "pong_raster_raw_sweep_kernel", "fused84": "pong_raster_gray", "fused84_newest": "pong_raster_gray", "fused84_f32": "pong_raster_gray",
            "fused84_f32_ref": "pong_gray_f32ref_kernel", "car": "car_obs_third_kernel", "car_fma": "car_obs_third_kernel", "tournament": "pong_policy_mfma_kernel",
            # protocol: the instance that draws the bound float32 frame stack + the uint8 observation (STACK, SF32)
            "protocol": "pong_raster_gray_env_kernel<3, false, 7, false, true, true,"}


def main():
    rnd, workloads = sys.argv[1], sys.argv[2:]
    for wl in workloads:
        src = os.path.join(ROOT, "gpurun_out", f"prof_{rnd}_{wl}")
        dst = os.path.join(ROOT, "profiles")
        if not os.path.isdir(src):
            print("missing", src)
            continue
        shutil.copy(os.path.join(src, "summary.txt"), os.path.join(dst, f"{rnd}_{wl}_summary.txt"))
        shutil.copy(os.path.join(src, "pmc_summary.json"), os.path.join(dst, f"{rnd}_{wl}_pmc_summary.json"))
        # the NEWEST trace only (tools/profile_gpu.sh wipes its output directory first; an older run's csv must never win)
        stats = sorted(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
        if stats:
            shutil.copy(stats[-1], os.path.join(dst, f"{rnd}_{wl}_kernel_stats.csv"))
        shutil.copy(os.path.join(src, "bench_trace.json"), os.path.join(dst, f"{rnd}_{wl}_bench_under_rocprof.json"))
        pmc = json.load(open(os.path.join(src, "pmc_summary.json")))
        fl = pmc.pop("__flops__", None)
        if fl and wl == "car":  # (car_fma: the same kernels with fused multiply-adds; flops_car.json stays the default arithmetic's)
            n_envs = 16384
            json.dump({"f32_flop_per_env_step": fl["f32_flop_per_step"] / n_envs, "f64_flop_per_env_step": fl["f64_flop_per_step"] / n_envs,
                       "valu_wave_insts_per_env_step": fl["valu_insts_per_step"] / n_envs,
                       "source": f"profiles/{rnd}_car_pmc_summary.json: rocprofv3 --pmc SQ_INSTS_VALU_{{ADD,MUL,FMA,TRANS}}_F32 over every car_* kernel of "
                                 f"{fl['steps']} steps at {n_envs} envs; 64 lanes per wave-instruction (upper bound), FMA = 2 FLOP"},
                      open(os.path.join(dst, "flops_car.json"), "w"), indent=1)
        if wl in ("car", "car_fma"):  # the raster also runs on a handful of envs for terminal frames: take the full-batch launches
            import csv
            for cname, sub in (("WRITE_SIZE", "pmc_write"), ("FETCH_SIZE", "pmc_fetch")):
                files = sorted(glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
                vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[-1]))
                        if r["Counter_Name"] == cname and DOMINANT[wl] in r["Kernel_Name"]]
                full = [v for v in vals if v > 0.5 * max(vals)]
                for name in pmc:
                    if DOMINANT[wl] in name:
                        pmc[name][cname] = sum(full) / len(full)
        for name, c in pmc.items():
            if DOMINANT[wl] in name and "template" not in name:
                w, f = c.get("WRITE_SIZE", 0.0) * 1024.0, c.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0
                json.dump({"kernel": name.split("(")[0], "hbm_bytes_per_launch": w + f, "write_bytes": w,
                           "fetch_bytes_corrected_x2": f,
                           "source": f"profiles/{rnd}_{wl}_pmc_summary.json (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE, separate "
                                     "passes; KiB units; FETCH_SIZE doubled per MI355X_MICROARCH.md)"},
                          open(os.path.join(dst, f"traffic_{wl}.json"), "w"), indent=1)
                print(wl, name.split("(")[0], "HBM bytes/launch", w + f)


if __name__ == "__main__":
    main()
