"""Pair the counter CSVs of `rocprofv3 --pmc ... -- python3 tools/probe_levels.py --tag T` with the launch schedule the probe
wrote (gpurun_out/levels/T.json) and print, per allocation of the feature pool, the mean duration and the mean of every
counter of its k_gather launches (and of the random-row probe's).

    python3 tools/levels_table.py gpurun_out/levels [tag ...] > profiles/r03_levels_counters.txt
"""
import collections
import csv
import glob
import json
import os
import sys


def kind_of(name):
    if "k_gather_heavy" in name:
        return None
    if "k_gather" in name:
        return "k_gather"
    if "k_probe_rows" in name:
        return "k_probe_rows"
    return None


def table(root, tag):
    js = os.path.join(root, tag + ".json")
    files = glob.glob(os.path.join(root, tag, "**", "*counter_collection.csv"), recursive=True)
    if not os.path.exists(js) or not files:
        return
    sched = json.load(open(js))["schedule"]
    disp = collections.OrderedDict()
    for f in files:
        for r in csv.DictReader(open(f)):
            k = kind_of(r["Kernel_Name"])
            if k is None:
                continue
            d = disp.setdefault(int(r["Dispatch_Id"]), {"kind": k, "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "c": {}})
            d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    order = [disp[k] for k in sorted(disp)]
    print(f"== {tag}: {len(order)} profiled dispatches, {len(sched)} scheduled launches")
    if len(order) != len(sched) or any(o["kind"] != s["kind"] for o, s in zip(order, sched)):
        print("   schedule and dispatches do not pair up; skipped")
        return
    groups = collections.OrderedDict()
    for o, s in zip(order, sched):
        if s.get("alloc", -1) < 0 or s.get("rep", 1) == 0:
            continue        # pre-pass and the warm-up repetition
        what = f"gather chunk {s['chunk']}" if s["kind"] == "k_gather" else f"rows window {s['window_gb']:g} GB"
        g = groups.setdefault((what, s["alloc"]), {"us": [], "c": collections.defaultdict(list)})
        g["us"].append(o["us"])
        for c, v in o["c"].items():
            g["c"][c].append(v)
    names = sorted({c for g in groups.values() for c in g["c"]})
    print("   " + f"{'what':24s} {'alloc':>5s} {'n':>3s} {'mean_us':>10s} " + " ".join(f"{c:>22s}" for c in names))
    for (what, al), g in groups.items():
        print("   " + f"{what:24s} {al:5d} {len(g['us']):3d} {sum(g['us']) / len(g['us']):10.1f} "
              + " ".join(f"{sum(g['c'][c]) / max(len(g['c'][c]), 1):22.6g}" for c in names))


if __name__ == "__main__":
    root = sys.argv[1]
    tags = sys.argv[2:] or sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(root, "*.json")))
    for t in tags:
        table(root, t)
