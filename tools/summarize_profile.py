"""Turn gpurun_out/prof_TAG (tools/profile.sh) into the committed summaries under profiles/:
  profiles/TAG_kernel_stats.csv   rocprofv3 --kernel-trace --stats per-kernel table (names shortened)
  profiles/TAG_counters.json      per kernel: launches, avg duration, HBM traffic per launch
                                  = (2 * FETCH_SIZE + WRITE_SIZE) KiB  (gfx950: FETCH_SIZE tallies 128-B
                                  requests at 64 B, MI355X_MICROARCH.md §HBM), MFMA busy fraction
usage: python tools/summarize_profile.py TAG"""
import csv, glob, json, os, re, sys, collections

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)

def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "")
    return name[:90]

def one(pattern):
    # a profile directory collects one file set per run (the file names carry the process id): take the newest
    fs = glob.glob(os.path.join(src, pattern), recursive=True)
    return max(fs, key=os.path.getmtime) if fs else None

out = {}
stats = one("trace/**/*kernel_stats.csv")
if stats:
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
            out.setdefault(short(r["Name"]), {}).update(calls=int(r["Calls"]), avg_us=round(float(r["AverageNs"]) / 1e3, 2),
                                                        pct=float(r["Percentage"]))

def counters(sub):
    f = one(sub + "/**/*counter_collection.csv")
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc

fetch, write, mfma = counters("pmc_fetch"), counters("pmc_write"), counters("pmc_mfma")
for k in set(fetch) | set(write) | set(mfma):
    d = out.setdefault(k, {})
    if "FETCH_SIZE" in fetch.get(k, {}):
        v = fetch[k]["FETCH_SIZE"]; d["fetch_kib_per_launch_raw"] = round(sum(v) / len(v), 1)
    if "WRITE_SIZE" in write.get(k, {}):
        v = write[k]["WRITE_SIZE"]; d["write_kib_per_launch"] = round(sum(v) / len(v), 1)
    if "fetch_kib_per_launch_raw" in d and "write_kib_per_launch" in d:
        d["hbm_bytes_per_launch"] = int((2 * d["fetch_kib_per_launch_raw"] + d["write_kib_per_launch"]) * 1024)
    m = mfma.get(k, {})
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m and sum(m["GRBM_GUI_ACTIVE"]) > 0:
        # MFMA_BUSY sums the 1024 SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs
        d["mfma_busy_frac"] = round(sum(m["SQ_VALU_MFMA_BUSY_CYCLES"]) / 1024.0 / (sum(m["GRBM_GUI_ACTIVE"]) / 8.0), 4)

summary = {"tag": tag, "kernels": out}
for fam_name, pat in (("gemm_family", "gemm_f32_kernel"), ("gemm_split_family", "gemm_x3_kernel"), ("gemm_split_small_family", "gemm_x3s_kernel"), ("gemm_h_family", "gemm_h_kernel")):
  gemm = {k: v for k, v in out.items() if pat in k}
  n = sum(v.get("calls", 0) for v in gemm.values())
  if n and all("hbm_bytes_per_launch" in v for v in gemm.values()):
    summary[fam_name] = {
        "launches": n,
        "avg_us": round(sum(v["avg_us"] * v["calls"] for v in gemm.values()) / n, 2),
        "hbm_bytes_per_launch": int(sum(v["hbm_bytes_per_launch"] * v["calls"] for v in gemm.values()) / n),
        "mfma_busy_frac": round(sum(v.get("mfma_busy_frac", 0) * v["avg_us"] * v["calls"] for v in gemm.values()) /
                                max(1e-9, sum(v["avg_us"] * v["calls"] for v in gemm.values())), 4)}
json.dump(summary, open(os.path.join(root, "profiles", f"{tag}_counters.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in summary.items() if k.endswith("_family")}, indent=1))
