#!/usr/bin/env python3
"""Distil rocprofv3 outputs under gpurun_out/ into the tracked profiles/ directory (round-tagged).

usage: tools/collect_profiles.py r1 c4 1000000000
  gpurun_out/prof_<round>/<wl>_kernel_stats.csv          -> profiles/<round>_<wl>_kernel_stats.csv
  gpurun_out/pmc_fetch|pmc_write/<wl>_counter_collection -> profiles/<round>_<wl>_pmc.csv + profiles/traffic.json
HBM bytes follow MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are in KiB, and on gfx950 FETCH_SIZE counts
exactly half of the bytes of a wide (16 B/lane) coalesced stream -> read bytes = 2 x FETCH_SIZE x 1024.
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, wl, rows = sys.argv[1], sys.argv[2], int(float(sys.argv[3]))
kern = {"c4": "k4_cmp_avg_by_group_main", "c2": "k2_region_count_main", "c3": "k3_flag_mapq_group_count_main",
        "c5": "k5_main", "c6": "k6_overlap_count_main"}[wl]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)
src = os.path.join(G, f"prof_{rnd}", f"{wl}_kernel_stats.csv")
if os.path.exists(src):
    shutil.copy(src, os.path.join(P, f"{rnd}_{wl}_kernel_stats.csv"))
bj = os.path.join(G, f"prof_{rnd}", f"bench_{wl}.json")
if os.path.exists(bj) and os.path.getsize(bj):
    line = open(bj).read().strip().splitlines()[-1]
    d = json.loads(line)
    if wl != "c4":
        d.pop("result", None)
    json.dump(d, open(os.path.join(P, f"{rnd}_bench_{wl}_1gpu.json"), "w"), indent=1)
for extra in ("bgzf_pipeline_kernel_stats.csv", "bam_pipeline_kernel_stats.csv", "fastq_pipeline_kernel_stats.csv", "vcfgz_end_to_end.log",
              "bam_end_to_end.log", "fastqgz_end_to_end.log"):
    e = os.path.join(G, f"prof_{rnd}", extra)
    if os.path.exists(e):
        shutil.copy(e, os.path.join(P, f"{rnd}_{extra}"))
vals = {}
out_rows = []
for name in ("fetch", "write"):
    f = os.path.join(G, f"pmc_{name}", f"{wl}_counter_collection.csv")
    if not os.path.exists(f):
        continue
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            out_rows.append({k: r[k] for k in ("Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "VGPR_Count",
                                               "SGPR_Count", "LDS_Block_Size", "Counter_Name", "Counter_Value")})
if out_rows:
    with open(os.path.join(P, f"{rnd}_{wl}_pmc.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(out_rows[0].keys()))
        w.writeheader()
        w.writerows(out_rows)
if "FETCH_SIZE" in vals:
    fetch = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"])
    write = sum(vals.get("WRITE_SIZE", [0])) / max(1, len(vals.get("WRITE_SIZE", [0])))
    tf = os.path.join(P, "traffic.json")
    t = json.load(open(tf)) if os.path.exists(tf) else {}
    t[wl] = {"round": rnd, "rows": rows, "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write,
             "correction": "read bytes = 2 x FETCH_SIZE x 1024 (gfx950 wide-stream undercount), write bytes = WRITE_SIZE x 1024",
             "hbm_bytes_per_launch": int(2 * fetch * 1024 + write * 1024)}
    json.dump(t, open(tf, "w"), indent=1)
    print(t[wl])
