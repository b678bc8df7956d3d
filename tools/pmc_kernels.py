#!/usr/bin/env python
"""Per-kernel SQ counter ratios from tools/pmc_kernels.sh (gpurun_out/pmck1, pmck2): where do the wave cycles go?"""
import collections, csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for d in ("pmck1", "pmck2"):
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", d, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_WAVE_CYCLES",): cnt[k] += 1
print("%-70s %9s %6s %6s %6s %6s %7s %7s %6s %6s" % ("kernel", "wavecyc/l", "act%", "wIns%", "wAny%", "mfma%", "valu/l", "mfma/l", "lds%", "vmem%"))
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    n = max(cnt[k], 1); wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    print("%-70s %9.0f %6.1f %6.1f %6.1f %6.1f %7.0f %7.0f %6.1f %6.1f" % (k, wc / n, 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc,
          100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 4 / wc, c.get("SQ_INSTS_VALU", 0) / n, c.get("SQ_INSTS_MFMA", 0) / n,
          100 * c.get("SQ_ACTIVE_INST_LDS", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_VMEM", 0) / wc))
