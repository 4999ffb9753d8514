#!/usr/bin/env python3
"""Condense rocprofv3 output under gpurun_out/ into the tracked summaries under profiles/.

usage: tools/summarize_prof.py TAG STATS_DIR [FETCH_DIR WRITE_DIR KEY]
  STATS_DIR : output of `rocprofv3 --kernel-trace --stats --output-format csv`
  FETCH_DIR / WRITE_DIR : outputs of separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes
  KEY : the runner's traffic_key() (kernel:precision:V:k<steps>:<threads>x<rows>:p<partitions>) under which bench.py
        looks the traffic up (profiles/traffic.json)
HBM traffic per launch = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes): on gfx950 FETCH_SIZE reports
half of the bytes of a coalesced streaming read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

def short(name):
    m = re.search(r"(\w*kernel\w*(<\w+>)?)", name)
    return m.group(1) if m else name[:60]


root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, stats_dir = sys.argv[1], sys.argv[2]
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)

rows = list(csv.DictReader(open(glob.glob(os.path.join(stats_dir, "**", "*_kernel_stats.csv"), recursive=True)[0])))
with open(os.path.join(out, tag + "_kernel_stats.csv"), "w") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys())
    w.writeheader()
    for r in rows:
        r["Name"] = short(r["Name"])
        w.writerow(r)
print(open(os.path.join(out, tag + "_kernel_stats.csv")).read())

if len(sys.argv) > 5:
    summary = {}
    for d, name in ((sys.argv[3], "FETCH_SIZE"), (sys.argv[4], "WRITE_SIZE")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0])):
            if r["Counter_Name"] == name:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            summary.setdefault(k, {})[name + "_KiB_avg"] = sum(v) / len(v)
            summary[k]["launches_" + name] = len(v)
    for k, v in summary.items():
        if "FETCH_SIZE_KiB_avg" in v and "WRITE_SIZE_KiB_avg" in v:
            v["hbm_bytes_per_launch"] = (2 * v["FETCH_SIZE_KiB_avg"] + v["WRITE_SIZE_KiB_avg"]) * 1024
    json.dump(summary, open(os.path.join(out, tag + "_pmc.json"), "w"), indent=1)
    print(json.dumps(summary, indent=1))
    key = sys.argv[5]
    kern = key.split(":")[0]
    tpath = os.path.join(out, "traffic.json")
    t = json.load(open(tpath)) if os.path.exists(tpath) else {}
    sys.path.insert(0, root)
    import fem_amd as fa          # the library the passes were run on travels with the tree: its digest ties the figure to the kernels
    for k, v in summary.items():
        if kern in k and "hbm_bytes_per_launch" in v:
            t[key] = {"hbm_bytes_per_launch": v["hbm_bytes_per_launch"], "source": "profiles/%s_pmc.json" % tag,
                      "kernel_build": fa.kernel_build()}
    json.dump(t, open(tpath, "w"), indent=1)
