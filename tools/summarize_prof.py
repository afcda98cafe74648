"""Summarise rocprofv3 CSV output (kernel_stats / kernel_trace / counter_collection) into a small text file
for profiles/.  Usage: python tools/summarize_prof.py <rocprof_dir> [<rocprof_dir> ...] > profiles/xxx.txt"""
import collections
import csv
import glob
import os
import sys


def short(name):
    for k in ("k_zero_call", "k_first_hit", "k_combine_parts", "k_gather_one", "k_gather", "k_viewtab", "k_build_cells", "k_block_dist",
              "k_build_near", "k_project_colors", "k_color_cells", "k_worklist", "k_stream_read", "k_aggregate_view_f16",
              "k_upsample_hwc", "k_chw_to_hwc", "k_voxel_coords", "k_scatter_occupancy", "k_nearest_voxel"):
        if k in name:
            return k
    return name[:60]


def from_rocpd(db):
    """rocprofv3's default output on ROCm 7.2 is a rocpd SQLite database: same two summaries from its views."""
    import sqlite3
    con = sqlite3.connect(db)
    views = {r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")}
    if "top_kernels" in views:
        rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
        if rows:
            print("-- kernel stats (rocprofv3 --kernel-trace --stats, rocpd top_kernels): name, calls, total_ms, avg_us, pct")
            for name, calls, tot, avg, pct in rows:
                if "k_" in name or float(pct) >= 1.0:
                    print(f"{short(name):20s} {int(calls):6d} {float(tot)/1e3:10.3f} {float(avg):10.1f} {float(pct):6.2f}")
    if "counters_collection" in views:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for name, counter, value in con.execute("select kernel_name, counter_name, value from counters_collection"):
            agg[short(name)][counter].append(float(value))
        if agg:
            print("-- counters (rocprofv3 --pmc, rocpd counters_collection): kernel, counter, launches, mean per launch")
            for k, cs in agg.items():
                if not k.startswith("k_"):
                    continue
                for c, v in cs.items():
                    print(f"{k:20s} {c:22s} {len(v):6d} {sum(v)/len(v):16.1f}")


for d in sys.argv[1:]:
    print(f"== {d}")
    for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
        from_rocpd(f)
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
