#!/bin/bash
# timeline of ONE single-frame extraction (B=1): start offset, duration, queue of every kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
rm -rf $O/prof_b1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_b1 -o b1 -- python $R/bench.py --batch 1 --streams 1 --steps 60 --warmup 5 --no-legs ) > $O/b1.json 2> $O/b1.err
python - <<'PY'
import csv, glob, os
f = glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/prof_b1/*kernel_trace.csv"))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_preproc" in r["Kernel_Name"]]
a, b = starts[40], starts[41]
t0 = int(rows[a]["Start_Timestamp"])
print("frame wall", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, "us")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:6.1f}  q{r.get('Queue_Id', '?'):>3}  {r['Kernel_Name'][:60]}")
PY
