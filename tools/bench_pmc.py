"""bench.py's reader and writers of the committed rocprofv3 PMC passes (profiles/<tag>_pmc_traffic.json, <tag>_r4_pmc.json).
Measurement plumbing only: imported by bench.py (the `roofline.traffic` field, `--write-pmc-json`, `--write-r4-pmc`)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "3d-semantic-segmentation_amd")

PMC_PROFILE = "r06_pmc_traffic.json"           # the round's committed counter passes (tools/profile_round.sh writes it)


def source_digest():
    """sha256 over the library's sources: a PMC profile only describes the kernels it was taken with."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(PKG, "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".h", ".hip")) or name == "Makefile":
            with open(os.path.join(csrc, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def pmc_traffic(workload, chunk, dtype):
    """HBM bytes per VIEW of a k_gather launch from the committed rocprofv3 PMC passes (profiles/<PMC_PROFILE>),
    corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE x2 for 16-B-per-lane streaming reads, KB units).
    None unless the profile was taken on this workload / views-per-call / dtype AND with these very kernel sources
    (the file records the digest of csrc/ it was measured on: a stale profile yields null, not a wrong number)."""
    path = os.path.join(ROOT, "profiles", PMC_PROFILE)
    try:
        with open(path) as f:
            prof = json.load(f)
    except OSError:
        return None
    run = prof.get("runs", {}).get(f"{workload}_{dtype}")
    # the profile's launches may hold a few views more or fewer than this run's (the per-view figure is what is used)
    if run is None or abs(run.get("views_per_call", 0) - chunk) > 4:
        return None
    if prof.get("source_digest") != source_digest():
        return None
    g = run["k_gather"]
    return (2.0 * g["FETCH_SIZE_KB_per_launch"] + g["WRITE_SIZE_KB_per_launch"]) * 1024 / run["views_per_call"]


PMC_RUNS = {   # key: (directory suffix of tools/profile_round.sh, views per launch of that pass = plan_calls' default for the leg)
    "R2_f32": ("f32", 60), "R2_f16": ("f16", 100), "R1_f32": ("R1", 50), "R2T_f32": ("R2T", 60), "A1_f32": ("A1", 54),
    "R2T_f16": ("R2T_f16", 100), "A1_f16": ("A1_f16", 108)}


def write_pmc_json(prof_dir, out_path):
    """profiles/<tag>_pmc_traffic.json from the counter CSVs of tools/profile_round.sh (FETCH_SIZE / WRITE_SIZE passes per
    leg: R2 fp32, R2 fp16, R1 fp32), stamped with the digest of the kernel sources they were measured on."""
    import collections
    import csv
    import glob
    runs = {}
    for key, (sfx, vpc) in PMC_RUNS.items():
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for kind in ("fetch", "write"):
            for f in glob.glob(os.path.join(prof_dir, f"{kind}_{sfx}", "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    name = r["Kernel_Name"]
                    k = "k_combine_parts" if "k_combine_parts" in name else "k_gather_one" if "k_gather_one" in name else "k_gather" if "k_gather" in name else \
                        "k_first_hit" if "k_first_hit" in name else None
                    if k:
                        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "k_gather" in per and per["k_gather"].get("FETCH_SIZE") and per["k_gather"].get("WRITE_SIZE"):
            # full launches only (two of them per pass; the pre-pass and the placement pass repeat them), at the views per
            # launch the default plan gives that leg
            runs[key] = {"views_per_call": vpc}
            runs[key].update({k: {"launches": len(v["FETCH_SIZE"]), "FETCH_SIZE_KB_per_launch": round(sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]), 1),
                                  "WRITE_SIZE_KB_per_launch": round(sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"])), 1)}
                              for k, v in per.items() if v.get("FETCH_SIZE") and v.get("WRITE_SIZE")})
    doc = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python3 bench.py --steps 1 --warmup 0 "
                     "--no-cpu-baseline --views 120 --chunk 60 | --views 200 --chunk 100 --dtype f16 | --workload R1 | --workload R2T | "
                     "--workload A1, MI355X (tools/profile_round.sh); the trajectory legs' launches differ (close-ups, misses): their "
                     "figures are means over the launches of one pass",
           "note": "gfx950: FETCH_SIZE counts 64 B per 128-B request for 16-B-per-lane streaming reads -> doubled by the reader "
                   "(MI355X_MICROARCH.md, HBM); WRITE_SIZE exact; units KB",
           "source_digest": source_digest(), "runs": runs}
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=1)
    return doc


def write_r4_pmc_json(prof_dir, out_path, views_per_call=1000):
    """profiles/<tag>_r4_pmc.json from the counter CSVs of tools/r4_round.sh (k_project_colors), stamped with the digest of the
    kernel sources: HBM-side bytes per launch and what the waves did with their cycles."""
    import collections
    import csv
    import glob
    c = collections.defaultdict(list)
    for f in glob.glob(os.path.join(prof_dir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_project_colors" in r["Kernel_Name"]:
                c[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in c.items()}
    summary = {"wave_cycles_parked_on_memory": round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 3),
               "wave_cycles_issue_stalled": round(m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], 3),
               # SQ_ACTIVE_INST_VALU counts quad-cycles summed over waves; GRBM_GUI_ACTIVE sums the busy cycles of the 8 XCDs
               "valu_busy_of_simd_cycles": round(m["SQ_ACTIVE_INST_VALU"] * 4 / (m["GRBM_GUI_ACTIVE"] / 8 * 1024), 3),
               "valu_instructions_per_voxel_view": round(m["SQ_INSTS_VALU"] * 64 / (500000.0 * views_per_call), 1),
               "l2_misses_per_launch": int(m["TCC_MISS_sum"]), "l2_hits_per_launch": int(m["TCC_HIT_sum"]), "waves": int(m["SQ_WAVES"])}
    if "TCP_TCC_READ_REQ_LATENCY_sum" in m and "TCP_TCC_READ_REQ_sum" in m:
        # The L1s (256 vector caches): line misses sent to the L2, their summed latency in cycles, and -- Little's law over the
        # launch's cycles (GRBM_GUI_ACTIVE sums the 8 XCDs) -- how many misses an L1 had in flight on average.  What the miss
        # queues sustained = misses in flight x 64 B / latency, per L1: bench.py quotes the R4 line against that.
        cycles = m["GRBM_GUI_ACTIVE"] / 8.0
        summary.update({"l1_line_lookups": int(m.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0)), "l1_misses_to_l2": int(m["TCP_TCC_READ_REQ_sum"]),
                        "l1_miss_latency_cycles": round(m["TCP_TCC_READ_REQ_LATENCY_sum"] / m["TCP_TCC_READ_REQ_sum"], 1),
                        "l1_misses_in_flight_per_l1": round(m["TCP_TCC_READ_REQ_LATENCY_sum"] / 256.0 / cycles, 2),
                        "l1_stalled_behind_pending_lines": round(m.get("TCP_PENDING_STALL_CYCLES_sum", 0.0) / 256.0 / cycles, 3),
                        "launch_cycles": int(cycles)})
    doc = {"source": "rocprofv3 --pmc passes of python3 bench.py --workload R4 --steps 1 --warmup 0 --no-cpu-baseline (tools/r4_round.sh), MI355X",
           "source_digest": source_digest(), "views_per_call": views_per_call,
           "FETCH_SIZE_KB_per_launch": round(m["FETCH_SIZE"], 1), "WRITE_SIZE_KB_per_launch": round(m["WRITE_SIZE"], 1), "summary": summary}
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=1)
    return doc
