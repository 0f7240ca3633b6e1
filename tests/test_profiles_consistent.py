"""profiles/: every committed ``<round>_<workload>_kernel_stats.csv`` must come from the same run as its ``_summary.txt``
(round 2 shipped csv files of an earlier run next to newer summaries): the top kernel and its average duration agree."""
import csv
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kernel_stats_csv_matches_summary():
    checked = 0
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[3456]_*_kernel_stats.csv"))):
        summ = f.replace("_kernel_stats.csv", "_summary.txt")
        assert os.path.exists(summ), summ
        rows = list(csv.DictReader(open(f)))
        top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
        text = open(summ).read()
        m = re.search(r"\{'Name': '" + re.escape(top["Name"]) + r"', 'Calls': '(\d+)', 'TotalDurationNs': '(\d+)', 'AverageNs': '([0-9.]+)'", text)
        assert m, (os.path.basename(f), top["Name"][:60])
        assert int(m.group(1)) == int(top["Calls"]), os.path.basename(f)
        assert abs(float(m.group(3)) - float(top["AverageNs"])) <= 0.02 * float(top["AverageNs"]), os.path.basename(f)
        checked += 1
    # (the files are regenerated at the end of a round; with none present there is nothing to compare)
    print("profiles checked:", checked)
