#!/usr/bin/env python
"""Per-kernel SQ counter ratios from tools/pmc_kernels.sh (gpurun_out/pmck1, pmck2): where do the wave cycles go?"""
import collections, csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else None          # with a tag: also profiles/<tag>_pmc_kernels<suffix>.json for tools/roofline_table.py
suffix = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for d in ("pmck1", "pmck2"):
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", d, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", ""))[:90]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_WAVE_CYCLES",): cnt[k] += 1
# kernel durations of the SAME run (pmck1 carries a kernel trace beside the counters)
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(ROOT, "gpurun_out", "pmck1", "*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        dur[re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", ""))[:90]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("%-90s %9s %6s %6s %6s %6s %7s %7s %6s %6s" % ("kernel", "wavecyc/l", "act%", "wIns%", "wAny%", "mfma%", "valu/l", "mfma/l", "lds%", "vmem%"))
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    n = max(cnt[k], 1); wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    print("%-90s %9.0f %6.1f %6.1f %6.1f %6.1f %7.0f %7.0f %6.1f %6.1f" % (k, wc / n, 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc,
          100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 4 / wc, c.get("SQ_INSTS_VALU", 0) / n, c.get("SQ_INSTS_MFMA", 0) / n,
          100 * c.get("SQ_ACTIVE_INST_LDS", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_VMEM", 0) / wc))

if tag:
    out = {"source": "tools/pmc_kernels.sh: rocprofv3 --pmc, two passes of 8 SQ counters over a serial bench step (one ctx, one stream)", "kernels": {}}
    for k, c in agg.items():
        n = max(cnt[k], 1); wc = c.get("SQ_WAVE_CYCLES", 0) or 1
        out["kernels"][k] = {"launches": n, "valu_per_launch": c.get("SQ_INSTS_VALU", 0) / n, "mfma_per_launch": c.get("SQ_INSTS_MFMA", 0) / n,
                             "mfma_cycles": 32.0 if "k_conv_mfma16" in k else 64.0,
                             # SQ_VALU_MFMA_BUSY_CYCLES counts busy SIMD-cycles (calibrated on k_sclk: 8192 MFMAs of 64 cycles per SIMD read 1024 x 524288); against the launch's
                             # duration in the same run x 1024 SIMDs x 2.4 GHz (the nominal clock: a lower real clock makes this a slight under-estimate)
                             "mfma_busy_pct": (100.0 * (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / n) / (sum(dur[k]) / len(dur[k]) * 1e-9 * 2.4e9 * 1024.0)) if dur.get(k) else None,
                             "avg_us_in_pmc_run": (sum(dur[k]) / len(dur[k]) / 1e3) if dur.get(k) else None,
                             "wave_cycles_per_launch": wc / n, "active_pct": 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc, "lds_bank_conflict_cycles_per_launch": c.get("SQ_LDS_BANK_CONFLICT", 0) / n}
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_kernels{suffix}.json"), "w"), indent=1)
