#!/bin/bash
# k_mnn_gemm_img launched back to back from an idle GPU: bracketing HIP events per window of 300 launches (tools/gemm_b2b.py) next to the rocprofv3
# --kernel-trace durations of the same launches -> gpurun_out/gemm_b2b.md (copied to profiles/r04_gemm_b2b.md)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
{
echo "# k_mnn_gemm_img, 4096 x 4096 x 64, launched back to back from an idle GPU (tools/gemm_b2b.sh, one MI355X)"
echo
echo "## without a profiler: two HIP events around each window of 300 launches (xfh_bench_mnn_gemm), 16 windows in a row"
echo '```'
REPS=16 python tools/gemm_b2b.py
echo '```'
echo
echo "## the same under rocprofv3 --kernel-trace --stats: per-dispatch durations of the same launches"
echo '```'
rm -rf $O/prof_gemm
( cd /tmp && REPS=16 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gemm -o g -- python $R/tools/gemm_b2b.py )
python - <<PY
import csv, glob
f = glob.glob("$O/prof_gemm/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "k_mnn_gemm_img" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
n = len(d)
print(f"rocprofv3: {n} launches of k_mnn_gemm_img")
for a in range(0, n, 320):
    w = d[a:a + 320]
    print(f"  launches {a:5d}..{a + len(w) - 1:5d}: average duration {sum(w) / len(w):6.2f} us = {2.0 * 4096 * 4096 * 64 / (sum(w) / len(w) * 1e-6) / 1e12 / 157.3:.3f} of 157.3 TFLOP/s")
s = d[-1600:]
print(f"settled (last 1600 launches): {sum(s) / len(s):.2f} us = {2.0 * 4096 * 4096 * 64 / (sum(s) / len(s) * 1e-6) / 1e12 / 157.3:.3f}")
for r in csv.DictReader(open(glob.glob("$O/prof_gemm/*kernel_stats.csv")[0])):
    if "k_mnn_gemm_img" in r["Name"]: print("kernel_stats.csv:", {k: r[k] for k in ("Calls", "AverageNs", "MinNs", "MaxNs") if k in r})
PY
echo '```'
} 2>&1 | grep -v "^W2026\|^E2026\|rocprofv3.*INFO" | tee $O/gemm_b2b.md
