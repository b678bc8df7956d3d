#!/bin/bash
# timeline of ONE single-frame extraction (B=1): start offset, duration, queue of every kernel (tools/b1_loop.py under rocprofv3 --kernel-trace)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
rm -rf $O/prof_b1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_b1 -o b1 -- python $R/tools/b1_loop.py $B1_ARGS ) > $O/b1.out 2> $O/b1.err
cat $O/b1.out
python - <<'PY'
import csv, glob, os
f = glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/prof_b1/*kernel_trace.csv"))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_preproc" in r["Kernel_Name"]]
a, b = starts[150], starts[151]
t0 = int(rows[a]["Start_Timestamp"])
print("frame wall", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, "us (under the profiler)")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    def dim(k):
        try: return int(r.get("Grid_Size_" + k, r.get("Grid_Size", 0))) // max(int(r.get("Workgroup_Size_" + k, r.get("Workgroup_Size", 1))), 1)
        except Exception: return 0
    wgs = max(dim("X"), 1) * max(dim("Y"), 1) * max(dim("Z"), 1)
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:6.1f}  q{r.get('Queue_Id', '?'):>3}  {wgs:5d} wg  {r['Kernel_Name'][:70]}")
PY
