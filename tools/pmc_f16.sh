#!/bin/bash
# SQ counters of the gather: fp16 C=512 vs fp32 C=256 (same row loads, different VALU work), serial phases
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
PMC="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
rm -rf gpurun_out/pmc_f16 gpurun_out/pmc_f32c256
rocprofv3 --pmc $PMC --kernel-trace --stats -d gpurun_out/pmc_f16 -o f16 --output-format csv -- python3 bench.py --dtype f16 --no-pipeline --pool-tries 1 --no-cpu-baseline --steps 1 --warmup 0 > gpurun_out/pmc_f16.log 2>&1 || exit 1
VOXPROJ_BENCH_C=256 rocprofv3 --pmc $PMC --kernel-trace --stats -d gpurun_out/pmc_f32c256 -o f32 --output-format csv -- python3 bench.py --dtype f32 --no-pipeline --pool 16 --pool-tries 1 --no-cpu-baseline --steps 1 --warmup 0 > gpurun_out/pmc_f32c256.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
for tag in ("pmc_f16", "pmc_f32c256"):
    for f in glob.glob(f"gpurun_out/{tag}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
        for k in acc:
            if "k_gather" in k and "heavy" not in k:
                print(tag, k, {c: (round(v / max(n[(k, c)], 1), 1)) for c, v in acc[k].items()}, "launches", max(n[(k, c)] for c in acc[k]))
PY
