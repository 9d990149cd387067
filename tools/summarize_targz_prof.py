#!/usr/bin/env python3
"""Condense a tools/profile_targz.sh output directory into tracked files under profiles/.

usage: tools/summarize_targz_prof.py gpurun_out/<dir> <tag>      (e.g. r02_targz_text)
Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, verbatim) and
profiles/<tag>_pmc.json: per kernel, the per-launch means of FETCH_SIZE / WRITE_SIZE (separate passes) and the HBM
bytes they stand for (FETCH_SIZE x 1024 x 2: the gfx950 correction of MI355X_MICROARCH.md; WRITE_SIZE x 1024)."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
ks = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
if ks:
    shutil.copy(ks[0], os.path.join(out, tag + "_kernel_stats.csv"))
res = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_valu"):
    for f in glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv")):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            m = re.search(r"(deflate_\w+_kernel|sha512_\w+_kernel)", r["Kernel_Name"])
            if not m:
                continue
            k = res.setdefault(m.group(1), {})
            k.setdefault("_launch", {"grid": r["Grid_Size"], "workgroup": r["Workgroup_Size"], "lds_bytes": r["LDS_Block_Size"],
                                     "vgpr": r["VGPR_Count"], "sgpr": r["SGPR_Count"]})
            agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for kn, cs in agg.items():
            for c, v in cs.items():
                res[kn][c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
for kn, k in res.items():
    k["_correction"] = "FETCH_SIZE*1024*2 (gfx950 reports half of a wide coalesced read; calibrated for 16 B/lane streaming only)"
    if "FETCH_SIZE" in k:
        k["hbm_read_bytes_per_launch"] = k["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2
    if "WRITE_SIZE" in k:
        k["hbm_write_bytes_per_launch"] = k["WRITE_SIZE"]["mean_per_launch"] * 1024
json.dump(res, open(os.path.join(out, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: {x: v[x] for x in v if x.startswith("hbm")} for k, v in res.items()}, indent=1))
