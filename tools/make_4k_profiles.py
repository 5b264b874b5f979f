#!/usr/bin/env python3
"""4K operating-point-4 pair: L2-miss traffic per kernel (FETCH_SIZE / WRITE_SIZE passes) -> profiles/<tag>_4k_pmc_traffic.json, and the
kernel stats of the tolerance mode -> profiles/<tag>_4k_op4_fast_kernel_stats.md.
usage: tools/make_4k_profiles.py <fetch_dir> <write_dir> <fast_stats_dir> <fast_log> <tag>"""
import collections, csv, glob, json, sys
fdir, wdir, kdir, klog, tag = sys.argv[1:6]
res = {}
for d, cn in ((fdir, "FETCH_SIZE"), (wdir, "WRITE_SIZE")):
    f = (glob.glob(d + "/*_counter_collection.csv") + glob.glob(d + "/*/*_counter_collection.csv"))[0]
    acc = collections.defaultdict(float); n = collections.Counter(); seen = set()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != cn or "fotg" not in r["Kernel_Name"]:
            continue
        key = "%s [grid %s]" % (r["Kernel_Name"].split("(")[0][:60], r["Grid_Size"])
        acc[key] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); n[key] += 1
    for k in acc:
        res.setdefault(k, {})[cn + "_KB_avg_per_launch"] = acc[k] / n[k]
        res[k]["launches_" + cn] = n[k]
for k, v in res.items():
    v["l2_miss_bytes_per_launch_corrected"] = (2 * v.get("FETCH_SIZE_KB_avg_per_launch", 0) + v.get("WRITE_SIZE_KB_avg_per_launch", 0)) * 1024
    v["hbm_bytes_per_launch_corrected"] = v["l2_miss_bytes_per_launch_corrected"]          # (the name earlier rounds used; same number)
out = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 tools/time_4k_op4.py",
       "workload": "BASELINE configs[3]: one 3840x2160 gray f32 pair, operating point 4; 7 calls",
       "semantics": "L2-miss bytes (requests that left an XCD's L2: Infinity-Cache hits included -- an upper bound of the HBM traffic)",
       "correction": "bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB counters x 1024); the x 2 is the guide's gfx950 correction for 16-byte-per-lane streaming reads (dword readers may be overstated on the read side)",
       "kernels": {k: res[k] for k in sorted(res, key=lambda k: -res[k]["l2_miss_bytes_per_launch_corrected"] * res[k].get("launches_FETCH_SIZE", 1))[:16]}}
json.dump(out, open("profiles/%s_4k_pmc_traffic.json" % tag, "w"), indent=1)
f = (glob.glob(kdir + "/*_kernel_stats.csv") + glob.glob(kdir + "/*/*_kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(f)))
line = [l.strip() for l in open(klog) if "per pair" in l]
with open("profiles/%s_4k_op4_fast_kernel_stats.md" % tag, "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/time_4k_op4.py --fast   (MI355X)\n")
    o.write("# BASELINE configs[3] in the tolerance mode (fotg_params::fast_math): one 3840x2160 gray f32 pair, operating point 4; 7 calls\n")
    o.write("".join("# %s (under the profiler)\n" % l for l in line) + "\n| kernel | calls | total ns | avg ns | % |\n|---|---|---|---|---|\n")
    for r in rows:
        if "fotg" in r["Name"] or "rocclr" in r["Name"]:
            o.write("| %s | %s | %s | %.0f | %s |\n" % (r["Name"][:110], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"]))
for k, v in list(out["kernels"].items())[:8]:
    print(k, round(v["l2_miss_bytes_per_launch_corrected"] / 1e6, 2), "MB x", v.get("launches_FETCH_SIZE"))
