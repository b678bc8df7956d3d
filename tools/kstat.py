"""per-kernel average duration from a rocprofv3 --kernel-trace csv: python tools/kstat.py trace.csv [min_grid_filter]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = acc.setdefault(n, [0, 0])
    a[0] += 1; a[1] += d
tot = sum(v[1] for v in acc.values())
for n, (c, d) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{d / c / 1e3:9.1f} us x {c:5d} = {d / 1e6:9.3f} ms ({100.0 * d / tot:5.1f} %)  {n[:150]}")
