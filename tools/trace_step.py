"""per-dispatch view of ONE train step from a rocprofv3 kernel trace (tools/profile.sh: gpurun_out/prof_TAG/trace):
kernel, grid, duration in launch order for the last step in the trace.   usage: python tools/trace_step.py TAG"""
import csv, glob, os, re, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fs = glob.glob(os.path.join(root, "gpurun_out", "prof_" + tag, "trace", "**", "*kernel_trace.csv"), recursive=True)
f = max(fs, key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    return n[:70]
# a step ends with adam_multi_kernel
ends = [i for i, r in enumerate(rows) if "adam_multi" in r["Kernel_Name"]]
if len(ends) < 2:
    sys.exit("fewer than two adam launches in the trace")
lo, hi = ends[-2] + 1, ends[-1] + 1
t0 = int(rows[lo]["Start_Timestamp"])
busy = 0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    g = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
    print(f"{(s - t0) / 1e3:9.1f}us +{(e - s) / 1e3:7.1f}us  wgs {g:6d} x {r['Workgroup_Size_X']:>4s}  lds {r.get('LDS_Block_Size', ''):>6s}  {short(r['Kernel_Name'])}")
print(f"step span {(int(rows[hi - 1]['End_Timestamp']) - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, {hi - lo} dispatches")
