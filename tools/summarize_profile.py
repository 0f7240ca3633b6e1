#!/usr/bin/env python3
"""Condenses a tools/profile_gpu.sh output directory into a small text summary:
per-kernel stats from the kernel trace and per-launch HBM bytes from the PMC passes
(FETCH_SIZE / WRITE_SIZE are in KiB-ish units of 1 KB; gfx950: FETCH_SIZE counts half the
bytes of wide coalesced reads -- see MI355X_MICROARCH.md "HBM")."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(root, pat):
    return sorted(glob.glob(os.path.join(root, "**", pat), recursive=True))


def main(out):
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    for f in sorted(find(os.path.join(out, "trace"), "*kernel_stats.csv"), key=os.path.getmtime)[-1:]:
        rows = list(csv.DictReader(open(f)))
        for r in rows[:12]:
            print({k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
    res = {}
    for cname in ("WRITE_SIZE", "FETCH_SIZE"):
        d = os.path.join(out, "pmc_write" if cname == "WRITE_SIZE" else "pmc_fetch")
        acc = defaultdict(list)
        for f in find(d, "*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == cname:
                    acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        print(f"== {cname} per launch (raw counter units) ==")
        for k, v in acc.items():
            print(k[:90], "launches", len(v), "mean", sum(v) / len(v), "min", min(v), "max", max(v))
            res.setdefault(k, {})[cname] = sum(v) / len(v)
    # counted vector FLOP (car only): wave-instruction counts x 64 lanes; FMA = 2 FLOP; summed over every kernel launch of the run
    fl = os.path.join(out, "pmc_flops")
    if os.path.isdir(fl):
        tot = defaultdict(float)
        launches = defaultdict(int)
        for f in find(fl, "*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "car_" in r["Kernel_Name"]:
                    tot[r["Counter_Name"]] += float(r["Counter_Value"])
                    if r["Counter_Name"] == "SQ_INSTS_VALU":
                        launches[r["Kernel_Name"].split("(")[0]] += 1
        steps = launches.get("crl::car_post_kernel", 0) or 1
        f32 = 64.0 * (tot["SQ_INSTS_VALU_ADD_F32"] + tot["SQ_INSTS_VALU_MUL_F32"] + tot["SQ_INSTS_VALU_TRANS_F32"] + 2 * tot["SQ_INSTS_VALU_FMA_F32"])
        f64 = 64.0 * (tot["SQ_INSTS_VALU_ADD_F64"] + tot["SQ_INSTS_VALU_MUL_F64"] + 2 * tot["SQ_INSTS_VALU_FMA_F64"])
        print("== counted vector FLOP (car kernels) ==", dict(tot), "steps", steps)
        res["__flops__"] = {"f32_flop_per_step": f32 / steps, "f64_flop_per_step": f64 / steps, "valu_insts_per_step": tot["SQ_INSTS_VALU"] / steps,
                            "steps": steps, "note": "64 x wave-instructions (masked-off lanes counted as active: an upper bound), FMA = 2"}
    json.dump(res, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
    for j in ("bench_trace.json", "bench_pmc_write.json"):
        p = os.path.join(out, j)
        if os.path.exists(p):
            print("==", j, "==")
            print(open(p).read().strip()[-1500:])


if __name__ == "__main__":
    main(sys.argv[1])
