#!/usr/bin/env python3
"""rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS passes -> profiles/<tag>_pmc_valu.json: instructions per
launch of the headline kernels (batch 64 x 1080p op-pt 2) and of the 4K op-pt 4 pair, with the derived figures DESIGN.md quotes
(VALU wave-instructions per four-patch LK iteration, VALU issue time at 1024 SIMDs x 0.6 G wave-instructions/s).
usage: tools/make_valu_profile.py <bench_pmc_dir> <4k_pmc_dir> <tag> [<bench_pmc_dir fast_math> <4k_pmc_dir fast_math>]"""
import collections, csv, glob, json, sys
bdir, kdir, tag = sys.argv[1:4]
fdirs = sys.argv[4:6]


def collect(d, want_grid=None):
    f = (glob.glob(d + "/*_counter_collection.csv") + glob.glob(d + "/*/*_counter_collection.csv"))[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if "fotg" not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size"]))
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        n[key].add(r["Dispatch_Id"])
    out = {}
    for key, c in acc.items():
        k = "%s [grid %d]" % key
        out[k] = {cn + "_per_launch": v / len(n[key]) for cn, v in c.items()}
        out[k]["launches"] = len(n[key])
        valu = out[k].get("SQ_INSTS_VALU_per_launch", 0.0)
        out[k]["valu_issue_us_at_1024_simds"] = valu / (1024 * 0.6e9) * 1e6
    return out


bench, k4 = collect(bdir), collect(kdir)
benchf, k4f = (collect(fdirs[0]), collect(fdirs[1])) if len(fdirs) == 2 else ({}, {})
notes = {}
for k, v in bench.items():
    if "lk_kernel<8, 1" in k and v.get("SQ_WAVES_per_launch", 0) > 30000:      # level 4: 510 patches x 64 pairs = 8160 waves ... per launch
        pass
for k, v in list(bench.items()) + list(k4.items()) + list(benchf.items()) + list(k4f.items()):
    if ("lk_kernel" in k or "lk_fast_kernel" in k) and v.get("SQ_WAVES_per_launch"):
        # a wave = four patches; every wave runs max_iter + 1 evaluations (12 + 1 at op-pt 2, 128 + 1 at op-pt 4)
        it = 129 if ("lk_kernel<12" in k or "lk_fast_kernel<12" in k) else 13
        ppw = 8 if ("lk_kernel<" in k and k.split("[")[0].rstrip().endswith(", 8>")) else 4          # lk_kernel<.., LPP = 8>: eight patches per wave
        v["patches_per_wave"] = ppw
        v["valu_per_wave_iteration"] = v["SQ_INSTS_VALU_per_launch"] / v["SQ_WAVES_per_launch"] / it
        v["valu_per_four_patch_iteration"] = v["valu_per_wave_iteration"] * 4 / ppw
json.dump({"command": "rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -- python3 bench.py --in-flight 1 --steps 3 "
                      "--warmup 1 --windows 1 --no-cpu-baseline --no-breakdown   (and: -- python3 tools/time_4k_op4.py)",
           "units": "wave-instructions per launch (SQ_INSTS_* count per wave); valu_per_four_patch_iteration = SQ_INSTS_VALU / SQ_WAVES / evaluations "
                    "(includes the template / Hessian / window set-up of the launch, spread over the evaluations)",
           "bench_batch64_1080p_op2": bench, "one_pair_4k_op4": k4,
           "fast_math_note": "the same two commands with --fast-math / --fast (fotg_params::fast_math, the tolerance mode: lk_fast_kernel, fused multiply-add solver updates)",
           "bench_batch64_1080p_op2_fast_math": benchf, "one_pair_4k_op4_fast_math": k4f}, open("profiles/%s_pmc_valu.json" % tag, "w"), indent=1)
for k, v in sorted(bench.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU_per_launch", 0))[:10]:
    print(k[:70].ljust(70), "VALU %.4g  waves %.4g  issue %.1f us" % (v.get("SQ_INSTS_VALU_per_launch", 0), v.get("SQ_WAVES_per_launch", 0), v["valu_issue_us_at_1024_simds"]), v.get("valu_per_four_patch_iteration", ""))
