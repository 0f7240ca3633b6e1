"""Which stream ends a CarRacing step, and how long the join takes (newest tools/car_timeline.sh trace): per step the period, the queue whose kernel
ended last, the last kernel end per queue (us from the step start) and the gap between that and the next car_step_kernel."""
import csv, glob, os
f = sorted(glob.glob("gpurun_out/car_timeline/t/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(f)) if "crl::" in r["Kernel_Name"]]
for r in rows: r["s"]=int(r["Start_Timestamp"]); r["e"]=int(r["End_Timestamp"]); r["n"]=r["Kernel_Name"].replace("crl::","").replace("void ","")[:24]
rows.sort(key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if "car_step_kernel" in r["n"]]
out=[]
for a, b in zip(starts[5:], starts[6:]):
    t0 = rows[a]["s"]; nxt = rows[b]["s"]
    seg = [r for r in rows[a:b] if "walk" not in r["n"]]
    last = {}
    for r in seg:
        q = r["Queue_Id"]
        if r["e"] <= nxt + 1000: last[q] = max(last.get(q, 0), r["e"])
    ends = {q: (e - t0) / 1e3 for q, e in last.items()}
    lastq = max(ends, key=ends.get)
    out.append((round((nxt - t0) / 1e3), lastq, {q: round(v) for q, v in sorted(ends.items())}, round((nxt - t0) / 1e3 - max(ends.values()))))
for o in out[-30:]: print(o)
import collections
print(collections.Counter(o[1] for o in out), "mean gap after the last kernel:", sum(o[3] for o in out) / len(out))
