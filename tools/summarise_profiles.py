#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/profile_round.sh (gpurun_out/prof_<tag>/) into the
committed summaries under profiles/:

    <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats table (hmv kernels only)
    <tag>_pmc_fetch.csv      FETCH_SIZE per dispatch (KB)
    <tag>_pmc_write.csv      WRITE_SIZE per dispatch (KB)
    <tag>_pmc_sq.csv         SQ counters, mean per launch per kernel
    <tag>_bench_n1.json      the bench line of the same build
    k3_traffic.json          HBM bytes per K3 launch = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction
                             of MI355X_MICROARCH.md: FETCH_SIZE counts 32 B per 64-B request of
                             16-B/lane streaming reads), read by bench.py as roofline.traffic

    python tools/summarise_profiles.py r01
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    if not hits:
        raise SystemExit(f"missing {pattern} under {src}")
    return hits[0]


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


# kernel stats
rows = list(csv.reader(open(one("stats/**/*kernel_stats.csv"))))
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    wr = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    wr.writerow(rows[0])
    for r in rows[1:]:
        if "hmv::" in r[0]:
            wr.writerow(r)

# PMC passes
per_kernel = {}
for cname, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    rd = csv.DictReader(open(one(f"{sub}/**/*counter_collection.csv")))
    acc = defaultdict(list)
    with open(os.path.join(dst, f"{tag}_pmc_{sub}.csv"), "w", newline="") as f:
        f.write("Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value_KB\n")
        for r in rd:
            if "hmv::" not in r["Kernel_Name"] or r["Counter_Name"] != cname:
                continue
            f.write('"%s","%s","%s","%s"\n' % (r["Dispatch_Id"], short(r["Kernel_Name"]), cname, r["Counter_Value"]))
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    per_kernel[cname] = {k: sum(v) / len(v) * 1024.0 for k, v in acc.items()}

try:
    rd = csv.DictReader(open(one("sq/**/*counter_collection.csv")))
    acc = defaultdict(list)
    for r in rd:
        if "hmv::" in r["Kernel_Name"]:
            acc[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    with open(os.path.join(dst, f"{tag}_pmc_sq.csv"), "w") as f:
        f.write("kernel,counter,mean_per_launch\n")
        for (k, c), v in sorted(acc.items()):
            f.write('"%s",%s,%.0f\n' % (k, c, sum(v) / len(v)))
except SystemExit as e:
    print("no SQ pass:", e)

k3 = "hmv::tf_inv_kernel<4, false>"
fetch, write = per_kernel["FETCH_SIZE"][k3], per_kernel["WRITE_SIZE"][k3]
W = 599
traffic = {
    "kernel": k3, "windows_per_launch": W,
    "FETCH_SIZE_bytes_raw": fetch, "WRITE_SIZE_bytes": write,
    "fetch_correction": "x2 (gfx950: FETCH_SIZE reports half of 16-B/lane streaming reads; MI355X_MICROARCH.md section HBM)",
    "hbm_bytes_per_launch": 2.0 * fetch + write,
    "algorithmic_bytes_per_launch": W * (64 * 64 * 256 * 8 + 64 * 64 * 8 * 8 + 64 * 256 * 8) + 8 * 256 * 16,
    "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, bench.py --steps 2 --warmup 1 "
              f"(profiles/{tag}_pmc_fetch.csv, {tag}_pmc_write.csv)",
}
json.dump(traffic, open(os.path.join(dst, "k3_traffic.json"), "w"), indent=1)

line = [l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")][-1]
res = json.loads(line)
res["roofline"]["traffic"] = traffic["hbm_bytes_per_launch"]
json.dump(res, open(os.path.join(dst, f"{tag}_bench_n1.json"), "w"), indent=1)
print(json.dumps({k: res[k] for k in ("value", "ms_per_step")}), res["roofline"]["frac"], traffic["hbm_bytes_per_launch"])
for r in rows[1:]:
    if "hmv::" in r[0]:
        print("%-50s calls %4s avg %10.1f us" % (short(r[0]), r[1], float(r[3]) / 1e3))
