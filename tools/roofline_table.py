#!/usr/bin/env python
"""profiles/<tag>_roofline_table.md: every kernel of one bench step (256 frames, one ctx, one stream) against its bound.

Input: the rocprofv3 kernel trace of `python bench.py --streams 1 --batch 256 --serial-branch` (gpurun_out/prof_serial/
*kernel_trace.csv, tools/gpu_round.sh): one ctx with the keypoint branch on the main stream, so no two kernels overlap and a
launch's duration is the kernel's own speed (the default bench runs four 64-frame ctx side by side, which is faster as a
whole -- profiles/<tag>_bench_sweep.md -- but stretches every single launch).
Launches are attributed to the batched steps by their grid (grid.z == B, or grid.x == B for the per-frame
kernels); algorithmic flops / bytes per launch come from the layer table: a convolution
reads its raw input map once, writes its raw output map once and does 2*H*W*Cout*Cin*k^2 flops per frame.
Peaks: f32 MFMA 157.3 TFLOP/s, HBM 8 TB/s (/opt/skills/guides/MI355X_MICROARCH.md)."""
import collections, csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = sys.argv[2] if len(sys.argv) > 2 else "prof_serial"          # gpurun_out/<src>/*kernel_trace.csv + gpurun_out/bench_<src>.json
suffix = sys.argv[3] if len(sys.argv) > 3 else ""                  # e.g. "_B64": profiles/<tag>_roofline_table<suffix>.md
f = glob.glob(os.path.join(ROOT, "gpurun_out", src, "*kernel_trace.csv"))[0]
_line = json.loads(open(os.path.join(ROOT, "gpurun_out", "bench_" + src + ".json")).read().strip().splitlines()[-1])
cfg = _line["config"]
MATCH_PAIRS = (_line.get("match", {}).get("batched") or {}).get("pairs", 8)
# per-kernel SQ counters of a serial step (tools/pmc_kernels.py writes profiles/<tag>_pmc_kernels.json): VALU instructions and MFMA-busy per launch
PMC = {}
try:
    PMC = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_pmc_kernels{suffix}.json")))["kernels"]
except Exception:
    pass
B, H, W = cfg["frames_per_gpu_per_step"] // cfg["sub_batches_in_flight"], cfg["height"] // 32 * 32, cfg["width"] // 32 * 32
PEAK_TF, PEAK_TB = 157.3, 8.0
# (cin, cout, k, stride) -> list of output sizes (h, w) of the layers that run on that instance
def conv_layers():
    h2, w2, h4, w4, h8, w8, h16, w16, h32, w32 = H // 2, W // 2, H // 4, W // 4, H // 8, W // 8, H // 16, W // 16, H // 32, W // 32
    return {(1, 4, 3, 1): (H, W), (4, 8, 3, 2): (h2, w2), (8, 8, 3, 1): (h2, w2), (8, 24, 3, 2): (h4, w4), (24, 24, 3, 1): (h4, w4),
            (24, 64, 3, 2): (h8, w8), (64, 64, 3, 1, "8"): (h8, w8), (64, 64, 1, 1): (h8, w8), (64, 64, 3, 2): (h16, w16),
            (64, 64, 3, 1, "16"): (h16, w16), (64, 128, 3, 2): (h32, w32), (128, 128, 3, 1): (h32, w32), (128, 64, 1, 1): (h32, w32)}
L = conv_layers()
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    gz, gx = int(r["Grid_Size_Z"]), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
    n = r["Kernel_Name"]
    batched = gz == B or (gx == B and any(k in n for k in ("k_bn_finalize", "k_select"))) or "k_conv_mfma_p" in n or "k_chain1x1" in n   # persistent kernels only run for B > 8
    if batched or any(k in n for k in ("k_mnn", "k_rownorm", "k_dist_mfma", "k_best2_csr", "k_distinctive_csr")):
        agg[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
steps = len(agg[[k for k in agg if "k_preproc" in k][0]])
rows = []
for n, d in agg.items():
    us = sum(d) / len(d) / 1e3
    MATCH = ("k_mnn", "k_rownorm", "k_dist_mfma", "k_best2_csr", "k_distinctive_csr")
    per_step = len(d) / steps if not any(k in n for k in MATCH) else 1
    flops = bytes_ = None; bound = "latency"
    m4 = re.match(r"void k_conv4_p<(\d+), (\d+), (\d+)", n)
    m = re.match(r"void k_conv_(mfma_p|mfma_t|mfma|direct)<(\d+), (\d+), (\d+), (\d+)(?:, (\d+), (\d+), (\d+), (\d+))?", n)
    if m4:                                            # 4x4x1 MFMA form of the 3x3 layers with few output channels
        cin, cout, st = int(m4.group(1)), int(m4.group(2)), int(m4.group(3))
        ho, wo = L[(cin, cout, 3, st)]
        flops = 2.0 * ho * wo * cout * cin * 9 * B
        bytes_ = 4.0 * B * (ho * st * wo * st * cin + ho * wo * cout)
        bound = "mfma" if flops / bytes_ > PEAK_TF / PEAK_TB else "hbm"
    elif "k_chain1x1" in n:                           # block_fusion.2 -> heatmap_head.0 in one pass: two 1x1 64->64 layers, feats written, raw map written
        ho, wo = L[(64, 64, 1, 1)]
        flops = 2 * 2.0 * ho * wo * 64 * 64 * B
        bytes_ = 4.0 * B * ho * wo * 64 * 3
        bound = "hbm"
    elif m:
        if m.group(1) != "direct":
            cin, cout, k, st, ww = int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(9))
        else:
            cin, cout, st, k, ww = int(m.group(2)), int(m.group(3)), int(m.group(4)), 3, 16
        key = (cin, cout, k, st)
        if key == (64, 64, 3, 1):
            key = key + (("8" if ww == 16 else "16"),)
        ho, wo = L[key]
        flops = 2.0 * ho * wo * cout * cin * k * k * B
        bytes_ = 4.0 * B * (ho * st * wo * st * cin + ho * wo * cout)
        if m.group(1) == "direct" and m.group(5) == "5":        # PRO_L0: block1.1 reads the 1-channel image and recomputes block1.0 (its 36 flops / pixel are counted here too)
            bytes_ = 4.0 * B * (H * W + ho * wo * cout)
            flops += 2.0 * H * W * 4 * 9 * B
        # the direct convolutions (block1) move 1.05 - 1.25 x their algorithmic bytes (profiles/pmc_traffic.json) at 30 - 45 % of the HBM peak: they are
        # bound by their per-pixel VALU work, not by traffic
        bound = "mfma" if m.group(1) != "direct" and flops / bytes_ > PEAK_TF / PEAK_TB else ("valu" if m.group(1) == "direct" else "hbm")
    elif "k_mnn_gemm_seg" in n:
        flops, bytes_, bound = 2.0 * 4096 * 4096 * 64 * MATCH_PAIRS, (1 + MATCH_PAIRS) * 4096 * 256.0, "mfma"
    elif "k_mnn_gemm" in n:
        flops, bytes_, bound = 2.0 * 4096 * 4096 * 64, 2 * 4096 * 256.0, "mfma"
    elif "k_dist_mfma" in n:                         # 4096 x 4096 int32 distance table (DescriptorDistance of every pair): 2.1 GFLOP of f32 MFMA (13.7 us) bound it before the 64 MB write (8.4 us) does
        flops, bytes_, bound = 2.0 * 4096 * 4096 * 64, 4096.0 * 4096 * 4 + 2 * 4096 * 256.0, "mfma"
    elif "k_mnn_post" in n:
        # bytes a pair's rows fetch from L2 / Infinity Cache (nothing of it is compulsory HBM traffic): per d1 row its row-key planes (16 x 8 B; 4 x 8 B behind
        # k_mnn_gemm_seg, which merges a panel's planes), its own row (256 B), MNN_CGROUP = 4 candidate d2 rows of 256 B, the column keys of the 4 candidates over 16 planes
        npairs = MATCH_PAIRS if "batch" in n else 1
        bytes_, bound = npairs * 4096.0 * ((32 if "batch" in n else 128) + 256 + 4 * 256 + 4 * 16 * 8), "cache"
    elif "k_best2_csr" in n: bytes_, bound = 4096.0 * 64 * (256 + 4) + 4096 * 256, "cache"      # 64 gathered 256-byte rows per query out of a 1 MB table: cache traffic, like k_mnn_post
    elif "k_distinctive_csr" in n: bytes_, bound = 4096.0 * 16 * (256 + 4), "cache"
    elif "k_block1_stats" in n: bytes_, bound, flops = 4.0 * B * H * W, "valu", 2.0 * H * W * 4 * 9 * B
    elif "k_act_pyramid" in n: bytes_, bound = 2 * 4.0 * B * 64 * ((H // 16) * (W // 16) + (H // 32) * (W // 32)), "hbm"      # x4 and x5 read and written once
    elif "k_feat_norm" in n: bytes_, bound = 4.0 * B * (H // 8) * (W // 8) * 65, "hbm"
    elif "k_heads_kp" in n:
        bytes_, bound = 4.0 * B * (H // 8 * (W // 8) * 64 + H * W), "valu"
        flops = 2.0 * B * (H // 8) * (W // 8) * 65 * 64
    elif "k_preproc" in n: bytes_, bound = B * H * W * 5.0, "valu"
    elif "k_norm_aux" in n: bytes_, bound = B * H * W * (4 + 4 + 0.25), "hbm"
    elif "k_b2in" in n: bytes_, bound = 4.0 * B * (H // 4) * (W // 4) * 24 * 2, "hbm"
    elif "k_fuse_in" in n: bytes_, bound = 4.0 * B * 64 * ((H // 8) * (W // 8) * 2 + (H // 16) * (W // 16) + (H // 32) * (W // 32)), "hbm"
    elif "k_feats_norm" in n: bytes_, bound = 4.0 * B * (H // 8) * (W // 8) * 64 * 2, "hbm"
    elif "k_heads_heat" in n: bytes_, bound = 4.0 * B * (H // 8) * (W // 8) * 65, "hbm"
    elif "k_nms_score" in n: bytes_, bound = 4.0 * B * H * W, "valu"
    # k_desc gathers 4 bilinear taps of 256 B per keypoint (4.2 MB per frame) from the normalised descriptor map: those reads are cache hits (the map is 1.23 MB per
    # frame and every line is fetched many times) -- round 5 priced them as HBM traffic and printed 101 % of the peak.  Compulsory HBM bytes: the map once + the record written
    elif "k_desc" in n: bytes_, bound = B * (4.0 * (H // 8) * (W // 8) * 64 + 4096.0 * 284), "hbm"
    elif "k_rownorm" in n: bytes_, bound = 2 * 4096 * 256.0 * 2, "hbm"
    if per_step < 0.9:                               # not a kernel of the timed step (a small-batch / eval()-mode instance that one of the bench legs launches): its bytes would be priced with the step's geometry
        continue
    rows.append((us * per_step, n, len(d), per_step, us, flops, bytes_, bound))
rows.sort(key=lambda r: -r[0])
MATCH = ("k_mnn", "k_rownorm", "k_dist_mfma", "k_best2_csr", "k_distinctive_csr")
tot = sum(r[0] for r in rows if not any(k in r[1] for k in MATCH))
out = os.path.join(ROOT, "profiles", f"{tag}_roofline_table{suffix}.md")
SCLK = 2.4e9
def pmc_of(n):
    for k, v in PMC.items():
        if n.startswith(k) or k.startswith(n[:len(k)]):
            return v
    return None
with open(out, "w") as o:
    o.write(f"# Every kernel of one bench step against its bound (B = {B} frames of {H}x{W}, one ctx, one stream, 1 x MI355X)\n\n"
            f"Source: rocprofv3 --kernel-trace of `python bench.py --streams 1 --batch {B} --serial-branch` (all kernels serial on one stream; tools/gpu_round.sh), launches with the batched grid only; {steps} steps.  The matcher kernels (k_mnn_*, k_rownorm_img, k_dist_mfma, k_best2_csr, k_distinctive_csr) are the 4096 x 4096 legs of the same run.\n"
            f"`alg` = algorithmic flops / HBM bytes per launch (raw input map read once + raw output map written once; no halo, no weights);\n"
            f"achieved = alg / average duration; % of the bound's peak (f32 MFMA {PEAK_TF} TFLOP/s, HBM {PEAK_TB} TB/s).  `latency` = per-frame single\n"
            f"workgroup or dependent-launch bound kernels (no meaningful roofline); `cache` = k_mnn_post[_batch], k_best2_csr, k_distinctive_csr: gathered rows / key planes out of tables that sit in L2 / Infinity Cache (no compulsory HBM bytes; the percentage is against the HBM rate for scale only).  Extraction kernels sum to {tot:.0f} us per step = {tot / B:.1f} us per frame.\n"
            f"`valu` = kernels whose traffic is within 1.05 - 1.25 x of algorithmic while they sit far below the HBM peak: bound by their vector work; for them (and as a second\n"
            f"figure for the MFMA kernels) `pipe` = share of the 1024 SIMDs' cycles the launch's plain VALU instructions (4 cycles each; SQ_INSTS_VALU minus the MFMAs it includes) and MFMAs (64 / 32 cycles) account for,\n"
            f"from the SQ counters of a serial step (tools/pmc_kernels.sh; VALU and f32 MFMA share one pipe on gfx950, profiles/{tag}_pipe_probe.log) at 2.4 GHz; `mfma busy` = SQ_VALU_MFMA_BUSY_CYCLES (busy SIMD-cycles, calibrated on a pure MFMA loop) / (launch duration in the counter run x 1024 SIMDs x 2.4 GHz).\n\n"
            "| kernel | launches/step | avg us | us/step | bound | alg GFLOP | alg MB | achieved | % of peak | pipe | mfma busy |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
    for tot_us, n, cnt, per, us, fl, by, bound in rows:
        nm = re.sub(r"\(.*", "", n.replace("void ", ""))
        if bound == "mfma": ach, pct = f"{fl / us / 1e6:.1f} TFLOP/s", f"{fl / us / 1e6 / PEAK_TF * 100:.0f} %"
        elif bound == "hbm":
            ach, pct = f"{by / us / 1e6:.2f} TB/s", f"{by / us / 1e6 / PEAK_TB * 100:.0f} %"
            assert by / us / 1e6 <= PEAK_TB, (nm, "an HBM-bound row above the HBM peak: its bytes are not all HBM traffic -- price the gathers as cache")
        elif bound == "valu": ach, pct = (f"{by / us / 1e6:.2f} TB/s" if by else "-"), (f"{by / us / 1e6 / PEAK_TB * 100:.0f} % of HBM" if by else "-")
        elif bound == "cache": ach, pct = f"{by / us / 1e6:.2f} TB/s from L2 / Infinity Cache", f"({by / us / 1e6 / PEAK_TB * 100:.0f} % of the HBM rate)"
        else: ach, pct = "-", "-"
        pm = pmc_of(nm)
        pipe = mb = "-"
        if pm:
            cyc = us * 1e-6 * SCLK * 1024.0
            pipe = f"{100.0 * ((pm['valu_per_launch'] - pm['mfma_per_launch']) * 4.0 + pm['mfma_per_launch'] * pm.get('mfma_cycles', 64.0)) / cyc:.0f} %"      # SQ_INSTS_VALU counts the MFMAs too
            mb = f"{pm['mfma_busy_pct']:.0f} %" if pm.get('mfma_busy_pct') is not None and pm['mfma_per_launch'] > 0 else "-"
        o.write(f"| `{nm}` | {per:.0f} | {us:.1f} | {tot_us:.1f} | {bound} | {fl / 1e9:.2f} | {by / 1e6:.1f} | {ach} | {pct} | {pipe} | {mb} |\n" if fl and by else
                f"| `{nm}` | {per:.0f} | {us:.1f} | {tot_us:.1f} | {bound} | {'-' if not fl else f'{fl / 1e9:.2f}'} | {'-' if not by else f'{by / 1e6:.1f}'} | {ach} | {pct} | {pipe} | {mb} |\n")
# ---- the same kernels inside the default bench's timed region (four ctx of 64 frames side by side): gpurun_out/prof/*kernel_trace.csv + gpurun_out/bench_prof.json
try:
    f2 = glob.glob(os.path.join(ROOT, "gpurun_out", "prof", "*kernel_trace.csv"))[0]
    l2 = json.loads(open(os.path.join(ROOT, "gpurun_out", "bench_prof.json")).read().strip().splitlines()[-1])
    c2 = l2["config"]; S2 = c2["sub_batches_in_flight"]; B2 = c2["frames_per_gpu_per_step"] // S2
    if not suffix and B2 != B:
        agg2 = collections.defaultdict(list)
        for r in csv.DictReader(open(f2)):
            gz, gx = int(r["Grid_Size_Z"]), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
            n = r["Kernel_Name"]
            if gz == B2 or (gx == B2 and any(k in n for k in ("k_bn_finalize", "k_select"))) or "k_conv_mfma_p" in n or "k_chain1x1" in n:
                agg2[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        nsub = len(agg2[[k for k in agg2 if "k_preproc" in k][0]])             # sub-batches traced (timed region + warm-up + the legs that run the same grid)
        ksum = sum(sum(d) for n, d in agg2.items() if len(d) >= 0.9 * nsub) / 1e3 / nsub * S2
        with open(out, "a") as o:
            o.write(f"\n## The same step as the bench runs it: {S2} ctx of {B2} frames side by side\n\n"
                    f"rocprofv3 --kernel-trace of the default `python bench.py` ({nsub} sub-batches of {B2} frames traced): the durations of one step's launches ({S2} sub-batches) add up to "
                    f"**{ksum:.0f} us**, the step takes **{l2['ms_per_step'] * 1e3:.0f} us** of wall time in that run ({l2['value']:.0f} frames/s): with {S2} streams a launch shares the CUs with the other "
                    f"sub-batches' kernels, so every single launch is longer than in the serial table above ({tot:.0f} us for {B} frames on one stream = {tot / B * B2 * S2:.0f} us per {B2 * S2} frames), "
                    f"and the overlap of the streams (memory-bound kernels of one sub-batch beside the MFMA-bound ones of another) is what brings the wall time below the serial sum.\n")
except Exception as e:       # noqa: BLE001
    print("in-region view skipped:", e)
print(open(out).read())
