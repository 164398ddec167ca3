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
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
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
    acc = defaultdict(list)
    for sub in ("sq", "sq2"):
        try:
            rd = csv.DictReader(open(one(f"{sub}/**/*counter_collection.csv")))
        except SystemExit:
            if sub == "sq":
                raise
            continue
        for r in rd:
            if "hmv::" in r["Kernel_Name"]:
                acc[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    with open(os.path.join(dst, f"{tag}_pmc_sq.csv"), "w") as f:
        f.write("kernel,counter,mean_per_launch\n")
        for (k, c), v in sorted(acc.items()):
            f.write('"%s",%s,%.0f\n' % (k, c, sum(v) / len(v)))
except SystemExit as e:
    print("no SQ pass:", e)

k3 = "hmv::tf_inv64_asm_kernel"        # the hand-scheduled 64-channel body (round 3); "hmv::tf_inv_kernel<4, false>" before
W = 599
# mean bytes per launch of every kernel: FETCH_SIZE x2 (gfx950 under-count of wide streaming reads) + WRITE_SIZE
table = {}
for k in sorted(set(per_kernel["FETCH_SIZE"]) | set(per_kernel["WRITE_SIZE"])):
    f_, w_ = per_kernel["FETCH_SIZE"].get(k, 0.0), per_kernel["WRITE_SIZE"].get(k, 0.0)
    table[k] = {"FETCH_SIZE_bytes_raw": f_, "WRITE_SIZE_bytes": w_, "hbm_bytes_per_launch": 2.0 * f_ + w_}
# the bench runs K3 in two forms (timed steps: normalisation inside; one check step: separate K4 pass); the PMC rows
# of the two are averaged per kernel name, so K3's figure is taken from the dispatches of the timed form only:
rd = csv.DictReader(open(one("write/**/*counter_collection.csv")))
k3w = [float(r["Counter_Value"]) * 1024.0 for r in rd if short(r["Kernel_Name"]) == k3 and r["Counter_Name"] == "WRITE_SIZE"]
rd = csv.DictReader(open(one("fetch/**/*counter_collection.csv")))
k3f = [float(r["Counter_Value"]) * 1024.0 for r in rd if short(r["Kernel_Name"]) == k3 and r["Counter_Name"] == "FETCH_SIZE"]
# timed form = the larger write volume (|H|^2 published + ffDTF written); 599-window launches only
full = [i for i, v in enumerate(k3w) if v > 0.8 * max(k3w)]
write = sum(k3w[i] for i in full) / len(full)
fetch = sum(k3f[i] for i in full) / len(full) if len(k3f) == len(k3w) else per_kernel["FETCH_SIZE"][k3]
traffic = {
    "round": tag,
    "kernel": k3 + " with the ffDTF normalisation inside", "windows_per_launch": W,
    "FETCH_SIZE_bytes_raw": fetch, "WRITE_SIZE_bytes": write,
    "fetch_correction": "x2 (gfx950: FETCH_SIZE reports half of 16-B/lane streaming reads; MI355X_MICROARCH.md section HBM)",
    "hbm_bytes_per_launch": 2.0 * fetch + write,
    "algorithmic_bytes_per_launch": W * (64 * 64 * 256 * 8 + 64 * 64 * 8 * 8 + 64 * 256 * 8) + 8 * 256 * 16,
    "note": "the launch now also does K4's work: |H|^2 is written once (write-through), read back once by the row "
            "normalisers and the ffDTF array written once: ~3x the algorithmic output bytes by construction",
    "per_kernel_mean_bytes_per_launch": table,
    "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, bench.py --steps 2 --warmup 1 "
              f"(profiles/{tag}_pmc_fetch.csv, {tag}_pmc_write.csv)",
}
json.dump(traffic, open(os.path.join(dst, "k3_traffic.json"), "w"), indent=1)

line = [l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")][-1]
res = json.loads(line)
res["roofline"]["traffic"] = traffic["hbm_bytes_per_launch"]
res["roofline"]["traffic_source"] = f"profiles/k3_traffic.json ({tag})"
json.dump(res, open(os.path.join(dst, f"{tag}_bench_n1.json"), "w"), indent=1)
print(json.dumps({k: res[k] for k in ("value", "ms_per_step")}), res["roofline"]["frac"], traffic["hbm_bytes_per_launch"])
for r in rows[1:]:
    if "hmv::" in r[0]:
        print("%-50s calls %4s avg %10.1f us" % (short(r[0]), r[1], float(r[3]) / 1e3))
try:
    srows = list(csv.reader(open(one("stats_spectra/**/*kernel_stats.csv"))))
    with open(os.path.join(dst, f"{tag}_kernel_stats_with_spectra.csv"), "w", newline="") as f:
        wr = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        wr.writerow(srows[0])
        for r in srows[1:]:
            if "hmv::" in r[0]:
                wr.writerow(r)
except SystemExit:
    pass
print("per-kernel HBM bytes per launch (2 x FETCH + WRITE):")
for k, v in table.items():
    print("  %-46s %10.1f MB" % (k, v["hbm_bytes_per_launch"] / 1e6))
