#!/usr/bin/env python3
"""Registers, LDS and occupancy of every kernel of libvoxproj (hipcc -Rpass-analysis=kernel-resource-usage), one line each.
python tools/kernel_resources.py [substring ...]"""
import os
import re
import subprocess
import sys

csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "3d-semantic-segmentation_amd", "csrc")
out = subprocess.run(["make", "-C", csrc, "asm"], capture_output=True, text=True)
text = out.stdout + out.stderr
cur, d = None, {}
for line in text.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        d[cur] = {}
    for k, short in (("VGPRs", "vgpr"), ("TotalSGPRs", "sgpr"), (r"Occupancy \[waves/SIMD\]", "waves/SIMD"), (r"LDS Size \[bytes/block\]", "lds"),
                     (r"ScratchSize \[bytes/lane\]", "scratch")):
        m = re.search(k + r": (\d+)", line)
        if m and cur:
            d[cur][short] = int(m.group(1))
want = sys.argv[1:]
filt = subprocess.run(["c++filt"] + list(d), capture_output=True, text=True).stdout.splitlines()
for name, (k, v) in zip(filt, d.items()):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)
    if not want or any(w in name for w in want):
        print(f"{name:60s} {v}")
