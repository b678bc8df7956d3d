"""Per hardware queue / stream summary of a rocprofv3 kernel trace: how the runtime spread the launches of one bench run
over its HSA queues, and how much of the time 1, 2, 3 ... kernels were executing at once.
usage: queue_view.py <kernel_trace.csv>"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
q = collections.Counter((r["Queue_Id"], r.get("Stream_Id", "?")) for r in rows)
print("launches per (queue, stream):", dict(q))
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
depth = 0; last = ev[0][0]; hist = collections.Counter()
for t, d in ev:
    hist[depth] += t - last; last = t; depth += d
tot = sum(hist.values())
print("time share by number of kernels executing:", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
print("span ms", (ev[-1][0] - ev[0][0]) / 1e6)
