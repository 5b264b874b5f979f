#!/usr/bin/env python3
"""Turns rocprofv3 outputs under gpurun_out/ into the committed summaries under profiles/.
usage: tools/make_profiles.py <stats_dir> <pmc_fetch_dir> <pmc_write_dir> <tag> [<4k_stats_dir> <4k_log> [<in_flight_stats_dir>]]"""
import collections, csv, glob, json, sys
stats_dir, fdir, wdir, tag = sys.argv[1:5]
f = (glob.glob(stats_dir + "/*_kernel_stats.csv") + glob.glob(stats_dir + "/*/*_kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(f)))
open("profiles/%s_bench_kernel_stats.csv" % tag, "w").write(open(f).read())
with open("profiles/%s_bench_kernel_stats.md" % tag, "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --in-flight 1 --steps 20 --warmup 3 --windows 3 --no-cpu-baseline --no-breakdown   (MI355X)\n")
    o.write("# ONLY the batch-64 launches of the headline path (63 steps): batch 64 x 1080p gray f32, op-pt 2 + refinement; averages are per launch of that grid\n")
    o.write("# (at::native kernels = synthetic input generation, outside the timed region)\n\n| kernel | calls | total ns | avg ns | % |\n|---|---|---|---|---|\n")
    for r in rows:
        o.write("| %s | %s | %s | %.0f | %s |\n" % (r["Name"][:110], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"]))
out = {}
for name, d in (("FETCH_SIZE", fdir), ("WRITE_SIZE", wdir)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open((glob.glob(d + "/*_counter_collection.csv") + glob.glob(d + "/*/*_counter_collection.csv"))[0])):
        if r["Counter_Name"] == name and "fotg" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out.setdefault(k, {})[name + "_KB_avg_per_launch"] = sum(v) / len(v)
        out[k]["launches_" + name] = len(v)
for k, d in out.items():
    d["hbm_bytes_per_launch_corrected"] = int(2 * d.get("FETCH_SIZE_KB_avg_per_launch", 0) * 1024 + d.get("WRITE_SIZE_KB_avg_per_launch", 0) * 1024)
json.dump({"command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --in-flight 1 --steps 3 --warmup 1 --windows 1 --no-cpu-baseline --no-breakdown ; same with --pmc WRITE_SIZE (separate passes)",
           "workload": "batch 64 x 1080p gray f32, op-pt 2 + refinement, MI355X",
           "correction": "bytes = 2*FETCH_SIZE[KB]*1024 + WRITE_SIZE[KB]*1024 (gfx950: FETCH_SIZE reports half of a wide streaming read; MI355X_MICROARCH.md, HBM)",
           "kernels": out}, open("profiles/%s_pmc_traffic.json" % tag, "w"), indent=1)
for r in rows[:12]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(5), ("%.1f" % (float(r["AverageNs"]) / 1000)).rjust(8), "us avg", r["Percentage"])
for k, d in out.items():
    if "pyr_base" in k: print(k, d)

if len(sys.argv) > 6:
    k4, log = sys.argv[5:7]
    f = (glob.glob(k4 + "/*_kernel_stats.csv") + glob.glob(k4 + "/*/*_kernel_stats.csv"))[0]
    rows = list(csv.DictReader(open(f)))
    line = [l.strip() for l in open(log) if "per pair" in l]
    with open("profiles/%s_4k_op4_kernel_stats.md" % tag, "w") as o:
        o.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/time_4k_op4.py   (MI355X)\n")
        o.write("# BASELINE configs[3]: one 3840x2160 gray f32 pair, operating point 4 (ps 12, 6 scales, 128 LK iterations, refinement); 7 calls\n")
        o.write("".join("# %s (under the profiler)\n" % l for l in line) + "\n| kernel | calls | total ns | avg ns | % |\n|---|---|---|---|---|\n")
        for r in rows:
            if "fotg" in r["Name"] or "rocclr" in r["Name"]:
                o.write("| %s | %s | %s | %.0f | %s |\n" % (r["Name"][:110], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"]))
    print(line[0] if line else "no 4K line")

if len(sys.argv) > 7:
    f = (glob.glob(sys.argv[7] + "/*_kernel_stats.csv") + glob.glob(sys.argv[7] + "/*/*_kernel_stats.csv"))[0]
    rows = list(csv.DictReader(open(f)))
    with open("profiles/%s_bench_inflight4_kernel_stats.md" % tag, "w") as o:
        o.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 3 --windows 3 --no-cpu-baseline --no-breakdown   (MI355X)\n")
        o.write("# the default run: FOUR batches in flight (fotg_pipe_*), kernels of different batches overlap, so a kernel's duration here includes\n")
        o.write("# what it loses to its neighbours; the one-batch-at-a-time durations are in %s_bench_kernel_stats.md\n\n| kernel | calls | total ns | avg ns | %% |\n|---|---|---|---|---|\n" % tag)
        for r in rows:
            if "fotg" in r["Name"]:
                o.write("| %s | %s | %s | %.0f | %s |\n" % (r["Name"][:110], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"]))
