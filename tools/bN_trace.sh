#!/bin/bash
# per-kernel cost breakdown at batch $1 (one stream): us per frame per kernel
B=${1:-32}
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
rm -rf $O/prof_bN
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_bN -o bN -- python $R/bench.py --batch $B --streams 1 --steps 30 --warmup 5 --match-iters 1 --cpu-frames 0 ) > $O/bN.json 2> $O/bN.err
B=$B python - <<'PY'
import csv, glob, collections, os
B = int(os.environ["B"])
f = glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/prof_bN/*kernel_trace.csv"))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_preproc" in r["Kernel_Name"]]
a, b = starts[6], starts[30]
seg = rows[a:b]; nstep = 24
wall = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / nstep
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    k = r["Kernel_Name"][:80]; agg[k][0] += 1; agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
busy = sum(v[1] for v in agg.values()) / nstep
print(f"B={B}: per step wall {wall/1e3:.1f} us, kernel busy {busy/1e3:.1f} us; per frame {wall/1e3/B:.2f} us")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{v[1]/nstep/1e3/B:7.2f} us/frame {v[0]/nstep:5.1f} launches avg {v[1]/v[0]/1e3:7.1f} us  {k}")
PY
