"""GPU idle time inside the train loop from a rocprofv3 --kernel-trace run: union of the kernel intervals against the span.
usage: python tools/gpu_idle.py DIR [skip_fraction]   (DIR holds *_kernel_trace.csv; the first skip_fraction of the span is dropped)"""
import csv, glob, sys
d = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t0, t1 = iv[0][0], max(e for _, e, _ in iv)
lo = t0 + (t1 - t0) * skip
iv = [x for x in iv if x[0] >= lo]
span = max(e for _, e, _ in iv) - iv[0][0]
busy = 0; cur_s, cur_e = iv[0][0], iv[0][1]
gaps = []
for s, e, n in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {100 * (1 - busy / span):.1f} %, kernels {len(iv)}, gaps {len(gaps)}")
import collections
by = collections.Counter(); cnt = collections.Counter()
for g, n in gaps:
    k = n.split("(")[0][:70]
    by[k] += g; cnt[k] += 1
print("idle time before a kernel, by the kernel that ends the gap (top 15):")
for k, v in by.most_common(15):
    print(f"  {v / 1e3:9.1f} us total  {cnt[k]:5d} gaps  avg {v / cnt[k] / 1e3:6.2f} us  {k}")
