"""Start / end (us from the step's first kernel) of every kernel of the last steady-state steps of the newest tools/car_timeline.sh trace."""
import csv, glob, os, sys
f = sorted(glob.glob("gpurun_out/car_timeline/t/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(f)) if "crl::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "car_step_kernel" in r["Kernel_Name"]]
periods = [(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3 for a, b in zip(starts, starts[1:])]
print("step periods (us), last 10:", [round(p) for p in periods[-10:]])
for si in starts[-4:-2]:
    t0 = int(rows[si]["Start_Timestamp"])
    print("---- step")
    for r in rows[si:]:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        if s > 1250:
            break
        print(f'{s:8.1f} {e:8.1f} q{r["Queue_Id"]} {r["Kernel_Name"].replace("crl::", "").replace("void ", "")[:30]}')
