#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh output directory into tracked files under profiles/.

usage: tools/summarize_prof.py gpurun_out/<dir> <tag> [workload]     (e.g. r02_pair_C2 C2)
Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, verbatim),
profiles/<tag>_pmc.json (per-launch means of every counter for the sha512 kernel, with
the gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md applied in `hbm_bytes_per_launch`)
and profiles/traffic_<kernel>.json, which bench.py reads for roofline.traffic.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "C2"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)
ks = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
if ks:
    shutil.copy(ks[0], os.path.join(out, tag + "_kernel_stats.csv"))
# The default command launches the hashing code two ways: ONE launch over the whole resident tree (the roofline object:
# 157 workgroups for 10 001 streams, kernel sha512_split_kernel<true>) and the staged batches of the end-to-end pass
# (4 096 streams = 64 workgroups each, the same code under the name sha512_pair_staged_kernel), so rocprofv3's --stats
# lists them apart; the per-launch trace, split by grid size, is kept as well.
kt = glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))
if kt:
    groups = collections.defaultdict(list)
    for r in csv.DictReader(open(kt[0])):
        if "sha512" in r["Kernel_Name"] or "deflate" in r["Kernel_Name"]:
            groups[(r["Kernel_Name"], r["Grid_Size_X"], r["Workgroup_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    with open(os.path.join(out, tag + "_kernel_by_grid.csv"), "w") as f:
        f.write("kernel,grid_size,workgroup_size,calls,avg_ms,min_ms,max_ms\n")
        for (k, g, w), v in sorted(groups.items(), key=lambda kv: -max(kv[1])):
            f.write('"%s",%s,%s,%d,%.4f,%.4f,%.4f\n' % (k, g, w, len(v), sum(v) / len(v), min(v), max(v)))
pmc = {}
kname = None
resident_grid = None
if kt:
    big = [(max(v), g) for (k, g, w), v in groups.items() if "sha512" in k and "staged" not in k]
    resident_grid = max(big)[1] if big else None
for d in ("pmc_fetch", "pmc_write", "pmc_sq"):
    for f in glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "sha512" in r["Kernel_Name"] and "staged" not in r["Kernel_Name"] and (resident_grid is None or r["Grid_Size"] == resident_grid):  # the resident launch only
                kn = r["Kernel_Name"]
                kname = ("sha512_split_kernel_true" if "<true>" in kn or "ILb1" in kn else "sha512_split_kernel_false") if "split" in kn else "sha512_wide_kernel"
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                pmc.setdefault("_launch", {"grid": r["Grid_Size"], "workgroup": r["Workgroup_Size"],
                                           "lds_bytes": r["LDS_Block_Size"], "vgpr": r["VGPR_Count"],
                                           "agpr": r["Accum_VGPR_Count"], "sgpr": r["SGPR_Count"]})
        for k, v in agg.items():
            pmc[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
if "FETCH_SIZE" in pmc:
    fetch = pmc["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2   # KiB units; gfx950 reports half of a wide coalesced read
    write = pmc.get("WRITE_SIZE", {"mean_per_launch": 0})["mean_per_launch"] * 1024
    pmc["hbm_bytes_per_launch"] = fetch + write
    pmc["_correction"] = "FETCH_SIZE*1024*2 + WRITE_SIZE*1024 (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reads 1/2 on gfx950)"
    bytes_per_launch = None
    bl = os.path.join(src, "bench_under_trace.log")
    if os.path.exists(bl):
        for l in open(bl, errors="replace"):
            if l.startswith("{"):
                bytes_per_launch = json.loads(l)["roofline"]["bytes_per_launch"]
    # bench.py quotes this figure only for the workload (and launch size) it was taken on
    json.dump({"hbm_bytes_per_launch": fetch + write, "source": tag + "_pmc.json", "workload": workload,
               "bytes_per_launch": bytes_per_launch},
              open(os.path.join(out, "traffic_%s.json" % kname), "w"))
json.dump(pmc, open(os.path.join(out, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
for name in ("bench_under_trace.log",):
    p = os.path.join(src, name)
    if os.path.exists(p):
        lines = [l for l in open(p, errors="replace") if l.startswith("{")]
        if lines:
            open(os.path.join(out, tag + "_bench_line.json"), "w").write(lines[-1])
print("wrote", sorted(f for f in os.listdir(out) if f.startswith(tag) or f.startswith("traffic")))
