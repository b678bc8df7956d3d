#!/bin/bash
# per-kernel breakdown of single-frame latency (B=1, one stream)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
rm -rf $O/prof_b1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_b1 -o b1 -- python $R/bench.py --batch 1 --streams 1 --steps 100 --warmup 5 --no-legs ) > $O/b1.json 2> $O/b1.err
python - <<'PY'
import csv, glob, collections, os
f = glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/prof_b1/*kernel_trace.csv"))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# take the timed 100 steps: find k_preproc launches
starts = [i for i, r in enumerate(rows) if "k_preproc" in r["Kernel_Name"]]
a, b = starts[10], starts[100]
seg = rows[a:b]; nstep = 90
wall = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / nstep
agg = collections.defaultdict(lambda: [0, 0]); gaps = 0; prev_end = None
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = r["Kernel_Name"][:70]; agg[k][0] += 1; agg[k][1] += e - s
    if prev_end is not None: gaps += max(0, s - prev_end)
    prev_end = e
busy = sum(v[1] for v in agg.values()) / nstep
print(f"per frame: wall {wall/1e3:.1f} us, kernel busy {busy/1e3:.1f} us, gaps {gaps/nstep/1e3:.1f} us, launches {len(seg)/nstep:.1f}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{v[1]/nstep/1e3:8.1f} us/frame  {v[0]/nstep:5.1f} launches  avg {v[1]/v[0]/1e3:7.2f} us  {k}")
PY
