#!/usr/bin/env python
"""gpurun_out/{prof,pmc_fetch,pmc_write} (rocprofv3 CSV) -> profiles/<tag>_*.csv + profiles/pmc_traffic.json.
HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE come from
separate --pmc passes, are in KiB, and on gfx950 FETCH_SIZE counts 128-B requests as 64 B -> doubled."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
G = os.path.join(ROOT, "gpurun_out"); P = os.path.join(ROOT, "profiles"); os.makedirs(P, exist_ok=True)

def kernel_trace(d):
    f = glob.glob(os.path.join(G, d, "*kernel_trace.csv"))
    rows = list(csv.DictReader(open(f[0]))) if f else []
    agg = collections.defaultdict(list)
    for r in rows:
        agg[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return agg

agg = kernel_trace("prof")
if agg:
    tot = sum(sum(v) for v in agg.values())
    with open(os.path.join(P, f"{tag}_bench_kernel_stats.csv"), "w") as f:
        w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, len(v), sum(v), round(sum(v) / len(v), 1), min(v), max(v), round(100.0 * sum(v) / tot, 2)])
    print("wrote kernel stats:", len(agg), "kernels")
for f in glob.glob(os.path.join(G, "prof", "*kernel_stats.csv")):
    os.replace(f, os.path.join(P, f"{tag}_rocprofv3_kernel_stats.csv"))

def pmc(d, name):
    f = glob.glob(os.path.join(G, d, "*counter_collection.csv"))
    out = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] == name:
                out[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return out

# ---- calibration (tools/pmc_calib.sh): counter value of kernels that move exactly 1 GiB per launch, per access width
CAL_BYTES = float(1 << 30)
CAL_MODES = {"k_calib<0>": ("read", 4), "k_calib<1>": ("read", 16), "k_calib<4>": ("read", 32), "k_calib<2>": ("write", 4), "k_calib<3>": ("write", 16), "k_calib<5>": ("write", 32)}
calib = {}
for (d, name) in (("calib_fetch", "FETCH_SIZE"), ("calib_write", "WRITE_SIZE")):
    for k, v in pmc(d, name).items():
        for tag_, (kind, width) in CAL_MODES.items():
            if tag_ in k:
                raw = sum(v) / len(v)
                e = calib.setdefault(f"{kind}_{width}B_per_lane", {})
                e[name + "_kib_raw"] = raw
                e[name + "_bytes_per_true_byte"] = raw * 1024.0 / CAL_BYTES
FF, WF = 2.0, 1.0          # factors that turn the raw KiB counters into bytes (defaults: the guide's gfx950 note; WRITE uncalibrated)
cal_note = "no calibration run found: FETCH x2 (MI355X_MICROARCH.md, HBM), WRITE x1 uncalibrated"
if calib.get("read_16B_per_lane", {}).get("FETCH_SIZE_bytes_per_true_byte") and calib.get("write_16B_per_lane", {}).get("WRITE_SIZE_bytes_per_true_byte"):
    FF = 1.0 / calib["read_16B_per_lane"]["FETCH_SIZE_bytes_per_true_byte"]
    WF = 1.0 / calib["write_16B_per_lane"]["WRITE_SIZE_bytes_per_true_byte"]
    cal_note = ("factors from tools/pmc_calib.sh on this box: kernels that read / write exactly 1 GiB per launch with 16 bytes per lane (the access width of "
                "nearly every load / store of the extraction); the other widths are listed in `calibration`")
if calib:
    print("calibration:", json.dumps(calib), "-> FETCH x%.3f WRITE x%.3f" % (FF, WF))

fetch, write = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
if fetch or write:
    per = {}
    for k in set(fetch) | set(write):
        fe = sum(fetch[k]) / len(fetch[k]) if fetch.get(k) else 0.0
        wr = sum(write[k]) / len(write[k]) if write.get(k) else 0.0
        per[k] = {"fetch_kib_raw": fe, "write_kib_raw": wr, "fetch_bytes": FF * fe * 1024.0, "write_bytes": WF * wr * 1024.0,
                  "hbm_bytes_per_launch": (FF * fe + WF * wr) * 1024.0, "launches": len(fetch.get(k, write.get(k, [])))}
    def pick(sub):
        for k, v in per.items():
            if sub in k:
                return v["hbm_bytes_per_launch"]
        return None
    B = None
    try:
        cfg = json.loads(open(os.path.join(G, "pmc_fetch.json")).read().strip().splitlines()[-1])["config"]
        B = cfg["frames_per_gpu_per_step"] // cfg["sub_batches_in_flight"]
    except Exception:
        pass
    # whole-step HBM traffic of the extraction: every kernel of the run except the matcher's, over the frames the run processed
    # (k_preproc launches 300 workgroups of 256 threads per VGA frame)
    def grid_frames(d):
        f = glob.glob(os.path.join(G, d, "*counter_collection.csv"))
        n = 0
        if f:
            seen = set()
            for r in csv.DictReader(open(f[0])):
                if "k_preproc" in r["Kernel_Name"] and r["Dispatch_Id"] not in seen:
                    seen.add(r["Dispatch_Id"]); n += int(r["Grid_Size"]) // 76800
        return n
    MATCH = ("k_mnn", "k_rownorm", "k_dist", "k_best2", "k_distinctive", "copyBuffer")
    ff, fw = grid_frames("pmc_fetch"), grid_frames("pmc_write")
    tot_f = sum(sum(v) for k, v in fetch.items() if not any(m in k for m in MATCH))
    tot_w = sum(sum(v) for k, v in write.items() if not any(m in k for m in MATCH))
    per_frame = (FF * tot_f / ff + WF * tot_w / fw) * 1024.0 if ff and fw else None
    js = {"extract_hbm_bytes_per_frame": per_frame, "extract_frames_counted": [ff, fw],
          "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; bytes = (fetch_factor * FETCH_SIZE + write_factor * WRITE_SIZE) * 1024 per launch; " + cal_note,
          "fetch_factor": FF, "write_factor": WF, "calibration": calib,
          "conv_batch": B, "conv_bytes_per_launch": pick("k_conv_mfma<64, 64, 3, 1, 4, 2, 1, 16, 1, 0, 32"),
          "gemm_bytes_per_launch": pick("k_mnn_gemm"), "per_kernel": per}
    json.dump(js, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
    print("wrote pmc_traffic.json; conv", js["conv_bytes_per_launch"], "gemm", js["gemm_bytes_per_launch"])
