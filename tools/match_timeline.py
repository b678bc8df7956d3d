"""GEMM / post durations and the gaps between them in the prepared-image match loop of a rocprofv3 kernel trace of
`bench.py --only-match-leg` (the last 200 k_mnn_post launches that directly follow a k_mnn_gemm_img launch)."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_mnn" in r["Kernel_Name"] or "k_rownorm" in r["Kernel_Name"]]
pairs = []
for i in range(1, len(seq) - 1):
    if "k_mnn_post" in seq[i][0] and "k_mnn_gemm" in seq[i - 1][0] and "k_mnn_gemm" in seq[i + 1][0]:
        g, p, n = seq[i - 1], seq[i], seq[i + 1]
        pairs.append((g[2] - g[1], p[1] - g[2], p[2] - p[1], n[1] - p[2], n[1] - g[1]))
pairs = pairs[-200:]
m = lambda k: sum(x[k] for x in pairs) / len(pairs) / 1e3
print("prepared loop, %d calls: gemm %.2f us | gap %.2f | post %.2f | gap to next gemm %.2f | period %.2f us" % (len(pairs), m(0), m(1), m(2), m(3), m(4)))
