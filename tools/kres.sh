#!/bin/bash
# per-kernel register / scratch / LDS use of one HIP source (clang's kernel-resource-usage remarks): tools/kres.sh kernels_conv.hip [extra flag] > out.txt
cd "$(dirname "$0")/../xfeatslam_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Rpass-analysis=kernel-resource-usage $2 -c "$1" -o /tmp/kres_$$.o 2>&1 |
  python3 -c '
import re, sys, subprocess
cur = None; rows = []
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = {"name": m.group(1)}; rows.append(cur); continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None: cur[key] = int(m.group(1))
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n")
for r, n in zip(rows, names):
    print("%-90s vgpr %3d agpr %3d scratch %4d occ %d" % (n[:90], r.get("vgpr", -1), r.get("agpr", -1), r.get("scratch", -1), r.get("occ", -1)))
'
rm -f /tmp/kres_$$.o
