#!/bin/bash
# per-kernel times of one serial 256-frame step for tools/ab/A.so and tools/ab/B.so on ONE box, side by side (kernels that differ by > 2 %)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cp xfeatslam_amd/libxfeat_hip.so /tmp/keep_abs.so
for v in A B; do
  cp tools/ab/$v.so xfeatslam_amd/libxfeat_hip.so
  rm -rf $O/prof_abs_$v
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_abs_$v -o t -- python $R/bench.py --streams 1 --batch ${AB_BATCH:-256} --serial-branch --no-legs --steps 6 --warmup 2 ) > $O/prof_abs_$v.log 2>&1
done
cp /tmp/keep_abs.so xfeatslam_amd/libxfeat_hip.so
python - <<'PY'
import csv, glob, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
def load(v):
    f = glob.glob(os.path.join(R, f"gpurun_out/prof_abs_{v}/*kernel_trace.csv"))[0]
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        a = acc.setdefault(r["Kernel_Name"].split("(")[0].replace("void ", ""), [0, 0]); a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return acc
A, B = load("A"), load("B")
ta, tb = sum(v[1] for v in A.values()), sum(v[1] for v in B.values())
print(f"total kernel time: A {ta / 1e6:.2f} ms, B {tb / 1e6:.2f} ms")
for n in sorted(set(A) | set(B), key=lambda n: -(A.get(n, [0, 0])[1] + B.get(n, [0, 0])[1])):
    a, b = A.get(n), B.get(n)
    if a and b:
        da, db = a[1] / a[0] / 1e3, b[1] / b[0] / 1e3
        if abs(da - db) > 0.02 * max(da, db): print(f"  A {da:8.1f} us  B {db:8.1f} us  x{a[0]:<4d} ({(db - da) * b[0] / 1e3:+7.2f} ms)  {n[:90]}")
    else:
        print(f"  only in {'A' if a else 'B'}: {(a or b)[1] / (a or b)[0] / 1e3:8.1f} us x{(a or b)[0]}  {n[:90]}")
PY
