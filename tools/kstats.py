"""print the per-kernel table of a rocprofv3 --kernel-trace --stats run: python tools/kstats.py DIR [rows] [steps]"""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 0
import os
rows = list(csv.DictReader(open(max(glob.glob(d + "/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime))))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms", round(tot / 1e6, 3), " per step" if steps else "", round(tot / 1e6 / steps, 4) if steps else "")
for r in rows[:n]:
    name = r["Name"].replace("void ", "")[:96]
    per = f"{int(r['Calls']) / steps:6.1f}/step {float(r['TotalDurationNs']) / 1e3 / steps:8.1f} us/step" if steps else ""
    print(f"{name:96s} {r['Calls']:>6s} {float(r['AverageNs']) / 1e3:9.1f} us {per}")
