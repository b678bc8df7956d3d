#!/bin/bash
# per-kernel average durations of the single-frame path for every build tools/ab/v_*.so (ablation / variant builds), on ONE box:
# each is swapped in for xfeatslam_amd/libxfeat_hip.so, `bench.py --batch 1` runs under rocprofv3 --kernel-trace, and the table
# lists the kernels whose name matches $AB_KERNELS (default: everything on the critical path)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cp xfeatslam_amd/libxfeat_hip.so /tmp/keep_ab.so
for v in tools/ab/v_*.so; do
  cp $v xfeatslam_amd/libxfeat_hip.so
  rm -rf $O/prof_ab
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_ab -o ab -- python $R/bench.py --batch 1 --streams 1 --steps 200 --warmup 20 --no-legs ) > $O/ab.json 2> $O/ab.err
  python - "$v" <<'PY'
import csv, glob, os, re, sys, json, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
f = glob.glob(os.path.join(R, "gpurun_out/prof_ab/*kernel_trace.csv"))[0]
pat = re.compile(os.environ.get("AB_KERNELS", "."))
d = collections.OrderedDict()
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
for r in rows[len(rows) // 2:]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    d.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
try: ms = json.loads(open(os.path.join(R, "gpurun_out/ab.json")).read().strip().splitlines()[-1])["ms_per_step"] * 1e3
except Exception: ms = float("nan")
print(f"== {sys.argv[1]}: {ms:.1f} us/frame (under the profiler)")
for n, v in d.items():
    if pat.search(n): print(f"   {sum(v) / len(v):7.2f} us x{len(v) // (len(rows) // 2 // max(1, len(d)) or 1):<3d} {n[:70]}")
PY
done
cp /tmp/keep_ab.so xfeatslam_amd/libxfeat_hip.so
