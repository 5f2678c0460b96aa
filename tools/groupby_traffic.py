#!/usr/bin/env python3
"""HBM traffic of the high-cardinality GROUP BY tier per 1e9-row step, from rocprofv3 PMC counters (FETCH_SIZE and WRITE_SIZE in
separate passes, each with --kernel-trace only; KiB; the 1 GiB calibration copy of bench.py --calib-copy gives bytes per count as
in bench.py's measure_traffic).  usage: tools/groupby_traffic.py <outdir> <groups> <dist> [ENV=VAL ...]"""
import csv, glob, os, subprocess, sys
out, groups, dist = os.path.abspath(sys.argv[1]), sys.argv[2], sys.argv[3]
env = dict(os.environ, TMPDIR="/tmp")
for kv in sys.argv[4:]:
    k, v = kv.split("=", 1)
    env[k] = v
os.makedirs(out, exist_ok=True)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = 3
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    d = os.path.join(out, "pmc_" + ctr)
    cmd = ["rocprofv3", "--pmc", ctr, "--kernel-trace", "-d", d, "-o", "p", "--output-format", "csv", "--", sys.executable,
           os.path.join(root, "bench.py"), "--workload", "c4", "--groups", groups, "--group-dist", dist, "--steps", str(steps), "--warmup", "1",
           "--no-cpu-baseline", "--no-extras", "--no-pmc", "--calib-copy"]
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=900)
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        sys.exit("no counter file: " + r.stderr[-2000:])
    rows = list(csv.DictReader(open(f[0])))
    cal = [float(x["Counter_Value"]) for x in rows if "copyBuffer" in x["Kernel_Name"] and x["Counter_Name"] == ctr and float(x["Counter_Value"]) > 0.25 * (1 << 20)]
    factor = (1 << 20) / (sum(cal) / len(cal)) if cal else (2.0 if ctr == "FETCH_SIZE" else 1.0)
    per = {}
    for x in rows:
        if x["Counter_Name"] != ctr or "exon::k4_" not in x["Kernel_Name"]:
            continue
        name = x["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
        per.setdefault(name, []).append(float(x["Counter_Value"]) * 1024 * factor)
    res[ctr] = (factor, per)
    subprocess.run(["rm", "-rf", d])
rows_step = 1e9
print(f"## groups {groups} {dist} {' '.join(sys.argv[4:])}: bytes per row of a 1e9-row step (all launches of the kernel summed, / (warmup + steps + gate))")
tot = 0.0
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    factor, per = res[ctr]
    for name, v in per.items():
        # launches of full steps only: drop the parity gate's smaller launches by taking the (warmup + steps) * 4 largest of main
        b = sum(sorted(v, reverse=True)[: (steps + 1) * 4]) / (steps + 1) / rows_step
        tot += b
        print(f"{ctr:10s} x{factor:.3f}  {name:48s} {b:7.3f} B/row  ({len(v)} launches)")
print(f"total {tot:.3f} B/row = {tot / 12.25:.3f} x the algorithmic 12.25 B/row")
