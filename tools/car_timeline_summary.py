"""Per-kernel count / median / max duration (us) of the newest tools/car_timeline.sh trace."""
import csv, glob, os, collections, sys
f = sorted(glob.glob("gpurun_out/car_timeline/t/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(f)) if "crl::" in r["Kernel_Name"]]
d = collections.defaultdict(list)
for r in rows:
    d[r["Kernel_Name"].replace("crl::", "")[:34]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{k:36s} n={len(v):4d} median={v2[len(v2)//2]:9.1f} max={max(v):9.1f}")
