"""Summarise rocprofv3 CSV output (kernel_stats / kernel_trace / counter_collection) into a small text file
for profiles/.  Usage: python tools/summarize_prof.py <rocprof_dir> [<rocprof_dir> ...] > profiles/xxx.txt"""
import collections
import csv
import glob
import os
import sys


def short(name):
    for k in ("k_first_hit", "k_gather_heavy", "k_gather", "k_viewtab", "k_build_cells", "k_block_dist",
              "k_build_near", "k_project_colors"):
        if k in name:
            return k
    return name[:60]


for d in sys.argv[1:]:
    print(f"== {d}")
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        print("-- kernel stats (rocprofv3 --kernel-trace --stats): name, calls, total_ms, avg_us, pct")
        for r in csv.DictReader(open(f)):
            print(f"{short(r['Name']):20s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:10.3f} "
                  f"{float(r['AverageNs'])/1e3:10.1f} {float(r['Percentage']):6.2f}")
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("-- counters (rocprofv3 --pmc): kernel, counter, launches, mean per launch")
        for k, cs in agg.items():
            if not k.startswith("k_"):
                continue
            for c, v in cs.items():
                print(f"{k:20s} {c:22s} {len(v):6d} {sum(v)/len(v):16.1f}")
