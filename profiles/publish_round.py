#!/usr/bin/env python3
"""Copies the evidence of one `python bench.py > gpurun_out/<tag>_bench.json; bash profiles/collect_pmc.sh <tag>`
run from gpurun_out/ (scratch) into profiles/ (tracked): the bench lines, the kernel stats, a PMC summary of this
run only, and traffic.json (HBM bytes per integrate launch = FETCH_SIZE x 2 + WRITE_SIZE, which bench.py reads for
roofline.traffic).   usage: python profiles/publish_round.py r02"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out", "pmc_" + tag)
prof = os.path.join(root, "profiles")
res = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("sq1", "fetch", "write", "tcc"):
    f = sorted(glob.glob(f"{out}/{d}/*/*counter_collection.csv"), key=os.path.getmtime)[-1]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if any(s in k for s in ("k_integrate", "k_pre_depth", "k_quality", "k_normal", "k_brick_clear", "k_skip", "k_window")):
            res[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in res.items()}
stats = sorted(glob.glob(out + "/stats/*/*kernel_stats.csv"), key=os.path.getmtime)[-1]
for r in csv.DictReader(open(stats)):
    if r["Name"] in summary:
        summary[r["Name"]]["avg_ns"] = float(r["AverageNs"])
        summary[r["Name"]]["calls"] = int(r["Calls"])
json.dump(summary, open(f"{prof}/{tag}_pmc_summary.json", "w"), indent=1, sort_keys=True)
shutil.copy(stats, f"{prof}/{tag}_kernel_stats.csv")
head = sorted(glob.glob(out + "/stats_headline/*/*kernel_stats.csv"), key=os.path.getmtime)
if head:   # the headline alone (bench.py --no-legs): the integrate kernel's average there is the one to hold against the bench's
    shutil.copy(head[-1], f"{prof}/{tag}_kernel_stats_headline_only.csv")
    shutil.copy(out + "/bench_stats_headline.json", f"{prof}/{tag}_bench_headline_only_under_rocprofv3.json")
    for r in csv.DictReader(open(head[-1])):
        if r["Name"] in summary and "k_integrate_tiled<" in r["Name"]:
            summary[r["Name"]]["avg_ns_headline_only"] = float(r["AverageNs"])
            summary[r["Name"]]["calls_headline_only"] = int(r["Calls"])
    json.dump(summary, open(f"{prof}/{tag}_pmc_summary.json", "w"), indent=1, sort_keys=True)
shutil.copy(out + "/bench_stats.json", f"{prof}/{tag}_bench_under_rocprofv3.json")
shutil.copy(os.path.join(root, "gpurun_out", tag + "_bench.json"), f"{prof}/{tag}_bench.json")
extra = os.path.join(root, "gpurun_out", tag + "_bench_extra.json")     # what the line moved out of itself (bench.py split_line)
if os.path.exists(extra):
    shutil.copy(extra, f"{prof}/{tag}_bench_extra.json")
line = json.load(open(f"{prof}/{tag}_bench.json"))
name = [k for k in summary if line["roofline"]["kernel"] in k][0]
c = summary[name]
traffic = {"4x512": {"kernel": name, "FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"],
                     "hbm_bytes_per_launch": int(round(c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024)),
                     "avg_launch_ns_rocprof": c.get("avg_ns_headline_only", c["avg_ns"]), "calls": c.get("calls_headline_only", c["calls"]),
                     "avg_launch_ns_rocprof_all_legs": c["avg_ns"], "calls_all_legs": c["calls"],
                     "correction": "FETCH_SIZE x2 (gfx950 reports half the bytes of 16-B/lane streaming reads, MI355X_MICROARCH.md "
                                   "'HBM'); WRITE_SIZE exact; separate --pmc passes (profiles/collect_pmc.sh)",
                     "source": f"profiles/{tag}_pmc_summary.json"}}
json.dump(traffic, open(f"{prof}/traffic.json", "w"), indent=1)
under = json.load(open(f"{prof}/{tag}_bench_under_rocprofv3.json"))
print("headline kernel", name)
print("  bench (HIP events, plain run): %.4f ms; under rocprofv3: events %.4f ms, rocprofv3 average %.4f ms over %d launches"
      % (line["roofline"]["avg_launch_ms"], under["roofline"]["avg_launch_ms"], c["avg_ns"] * 1e-6, c["calls"]))
if "avg_ns_headline_only" in c:
    ho = json.load(open(f"{prof}/{tag}_bench_headline_only_under_rocprofv3.json"))
    print("  headline alone (--no-legs) under rocprofv3: events %.4f ms, rocprofv3 average %.4f ms over %d launches"
          % (ho["roofline"]["avg_launch_ms"], c["avg_ns_headline_only"] * 1e-6, c["calls_headline_only"]))
print("  PMC traffic %.3f GB per launch (algorithmic %.3f GB)" % (traffic["4x512"]["hbm_bytes_per_launch"] / 1e9,
                                                                   line["roofline"]["bytes_per_launch"] / 1e9))
for k, v in sorted(summary.items()):
    if "FETCH_SIZE" in v and "avg_ns" in v:
        print("  %-78s %8.1f us  %7.3f GB  %6.1f M VALU" % (k[:78], v["avg_ns"] / 1e3,
              (v["FETCH_SIZE"] * 2 + v["WRITE_SIZE"]) * 1024 / 1e9, v.get("SQ_INSTS_VALU", 0) / 1e6))
