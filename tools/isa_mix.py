#!/usr/bin/env python
"""Instruction mix of every kernel of the serial roofline table, from the disassembly of the SHIPPED library.

    python tools/isa_mix.py [--pmc profiles/r06_pmc_kernels.json] [--table profiles/r06_roofline_table.md] > profiles/r06_isa_mix.md

What it does
  1. pulls the gfx950 code objects out of xfeatslam_amd/libxfeat_hip.so (the .hip_fatbin section holds one clang offload bundle per translation unit) and
     disassembles them with llvm-objdump --symbolize-operands;
  2. per kernel: finds the loops (a branch to a label at a lower address), the nesting depth of every instruction, and classifies every instruction:
       mfma       v_mfma_*                                            (64 pipe cycles for 32x32x2 f32, 32 for 16x16x4 f32 -- f32 MFMA issues on the vector pipe)
       fma32      v_fma / v_fmac / v_mul / v_add / v_sub / v_mad on f32, packed or not (v_pk_fma_f32 counts as ONE instruction: 2 lanes of work in 4 cycles... at pk rate)
       f64        any *_f64 VALU instruction                           (8 or 16 cycles: a quarter / an eighth of the f32 rate)
       cvt/trans  v_cvt_*, v_exp/log/rcp/rsq/sqrt/sin/cos, v_fract/floor/rndne ...   (transcendentals: 16 cycles)
       cmp/sel    v_cmp*, v_cndmask, v_max/min/med3 on any type        (the arg-max / border / ReLU work)
       int/addr   integer add / mul / shift / logic / v_mov / v_perm / v_readlane ...: address arithmetic and data movement between registers
       lds        ds_*            vmem  global_* / buffer_* / flat_* / scratch_*            salu  s_* (not the waits / nops / barriers)      wait  s_waitcnt / s_nop / s_barrier / s_sleep
  3. joins the static counts with the DYNAMIC totals of the SQ counters (tools/pmc_kernels.sh: SQ_INSTS_VALU, SQ_INSTS_MFMA ... per launch of a serial
     256-frame step): for an MFMA kernel the trip count of its K loop follows from dyn MFMA / static MFMA of the innermost MFMA loop, which splits the dynamic VALU
     count into "inside the K loop" (static mix x trips) and "outside" (prologue + epilogue: staging, statistics, stores); for a VALU kernel the static mix of
     its hottest loop (or of the whole kernel when it is unrolled flat) is scaled to the dynamic VALU count.
  4. prices the pipe: cycles = 4 x plain VALU + 64 (32) x MFMA, over 1024 SIMDs x 2.4 GHz x the measured duration = the share of the vector pipe the launch's own
     instructions account for; what is left is issue stalls / waits (memory, LDS, barriers, dependent-MFMA latency).

Static analysis has limits, stated where they matter: predicated-off instructions and branches not taken still count statically; trip counts of VALU kernels
with several sibling loops are taken from the counters' total, not measured per loop."""
from __future__ import annotations

import argparse
import json
import os
import re
import struct
import subprocess
import sys
from collections import Counter, defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
CXXFILT = "/usr/bin/c++filt"
CLASSES = ["mfma", "fma32", "f64", "cvt/trans", "cmp/sel", "int/addr", "lds", "vmem", "salu", "wait", "branch"]


def code_objects(so_path):
    """gfx950 ELF images inside the library's .hip_fatbin section"""
    tmp = "/tmp/isa_mix"
    os.makedirs(tmp, exist_ok=True)
    fat = os.path.join(tmp, "fatbin")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so_path, fat])
    d = open(fat, "rb").read()
    mag, out, pos = b"__CLANG_OFFLOAD_BUNDLE__", [], 0
    while True:
        i = d.find(mag, pos)
        if i < 0:
            break
        n = struct.unpack_from("<Q", d, i + 24)[0]
        o = i + 32
        for _ in range(n):
            off, size, ts = struct.unpack_from("<QQQ", d, o); o += 24
            triple = d[o:o + ts].decode(); o += ts
            if "gfx950" in triple and size:
                p = os.path.join(tmp, f"co{len(out)}.elf")
                open(p, "wb").write(d[i + off:i + off + size])
                out.append(p)
        pos = i + 24
    return out


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_setprio", "s_sethalt", "s_endpgm", "s_code_end")):
            return "wait"
        if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_call")):
            return "branch"
        return "salu"
    if not op.startswith("v_"):
        return "salu"
    if "_f64" in op:
        return "f64"
    if re.match(r"v_(cvt|exp|log|rcp|rsq|sqrt|sin|cos|fract|floor|ceil|trunc|rndne|frexp|ldexp)", op):
        return "cvt/trans"
    if re.match(r"v_(cmp|cndmask|max|min|med3|cmpx)", op):
        return "cmp/sel"
    if re.match(r"v_(pk_)?(fma|fmac|mul|add|sub|subrev|mad|mac)_(f32|f16|legacy_f32)", op) or re.match(r"v_pk_(fma|mul|add)_f32", op) or op.startswith("v_dot"):
        return "fma32"
    return "int/addr"


def parse(elf):
    """{kernel symbol: [(addr, opcode, class, target label or None)]} and {label: addr} per kernel"""
    txt = subprocess.run([OBJDUMP, "-d", "--symbolize-operands", elf], capture_output=True, text=True, check=True).stdout
    kernels, cur, labels = {}, None, {}
    lab_of = {}
    for line in txt.splitlines():
        m = re.match(r"^([0-9a-f]{16}) <(.+)>:$", line)
        if m:
            addr, name = int(m.group(1), 16), m.group(2)
            if re.fullmatch(r"L\d+", name):
                if cur is not None:
                    lab_of[cur][name] = addr
            else:
                cur = name
                kernels[cur] = []
                lab_of[cur] = {}
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]{12}):", line)
        if m and cur is not None:
            op, args, addr = m.group(1), m.group(2), int(m.group(3), 16)
            tgt = None
            if op.startswith(("s_cbranch", "s_branch")):
                t = re.search(r"\b(L\d+)\b", args)
                tgt = t.group(1) if t else None
            kernels[cur].append((addr, op, classify(op), tgt, args))
    return kernels, lab_of


def analyse(ins, labels):
    """loops = [(start addr, end addr)], depth per instruction, counts per depth"""
    loops = []
    for addr, op, cls, tgt, _ in ins:
        if tgt and tgt in labels and labels[tgt] <= addr:
            loops.append((labels[tgt], addr))
    depth = []
    for addr, *_ in ins:
        depth.append(sum(1 for a, b in loops if a <= addr <= b))
    return loops, depth


def demangle(names):
    out = subprocess.run([CXXFILT], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    return dict(zip(names, out))


def short(dem):
    """'void k_conv_mfma<64, 64, ...>(args)' -> 'k_conv_mfma<64, 64, ...>' (the spelling of the roofline table / the counter files)"""
    s = re.sub(r"^void ", "", dem)
    depth, cut = 0, len(s)
    for i, ch in enumerate(s):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    s = s[:cut]
    return s.replace("(bool)0", "false").replace("(bool)1", "true").replace("(XfhKernel)", "")


def mfma_cycles(op):
    return 32 if "16x16x4" in op else 64 if "32x32x2" in op else 8 if "4x4x1" in op else 64


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--so", default=os.path.join(ROOT, "xfeatslam_amd", "libxfeat_hip.so"))
    ap.add_argument("--pmc", default=None, help="profiles/rNN_pmc_kernels.json (serial B = 256 step); default: the newest one")
    ap.add_argument("--table", default=None, help="profiles/rNN_roofline_table.md: kernel order, launches per step and durations")
    ap.add_argument("--frames", type=int, default=256)
    a = ap.parse_args()
    prof = os.path.join(ROOT, "profiles")
    if a.pmc is None:
        c = sorted(f for f in os.listdir(prof) if re.fullmatch(r"r\d+_pmc_kernels\.json", f))
        a.pmc = os.path.join(prof, c[-1])
    if a.table is None:
        c = sorted(f for f in os.listdir(prof) if re.fullmatch(r"r\d+_roofline_table\.md", f))
        a.table = os.path.join(prof, c[-1])
    pmc = json.load(open(a.pmc))["kernels"]
    # roofline table rows: | `kernel` | launches/step | avg us | us/step | bound | ...
    rows = []
    for line in open(a.table):
        m = re.match(r"^\| `([^`]+)` \| (\d+) \| ([\d.]+) \| ([\d.]+) \| (\w+) ", line)
        if m:
            rows.append((m.group(1), int(m.group(2)), float(m.group(3)), float(m.group(4)), m.group(5)))
        if line.startswith("## "):
            break
    kern, labs = {}, {}
    for elf in code_objects(a.so):
        k, l = parse(elf)
        kern.update(k); labs.update(l)
    dem = demangle(list(kern))
    by_short = defaultdict(list)
    for sym in kern:
        by_short[short(dem[sym])].append(sym)

    print(f"# Instruction mix per kernel, from the disassembly of the shipped `libxfeat_hip.so` (tools/isa_mix.py)\n")
    print(f"Static counts: `llvm-objdump -d` of the gfx950 code objects inside `xfeatslam_amd/libxfeat_hip.so`; dynamic totals: `{os.path.relpath(a.pmc, ROOT)}` (SQ counters of a serial "
          f"{a.frames}-frame step); launches and durations: `{os.path.relpath(a.table, ROOT)}`.  Classes and method: the docstring of `tools/isa_mix.py`.\n")
    print("`VALU/out` = dynamic plain-VALU instructions per output element of the layer (lane-instructions: wave instructions x 64 / outputs, i.e. what one output costs on the vector pipe "
          "besides its MFMAs); `pipe` = (4 x VALU (8 for the kernel's static fp64 share) + 64 | 32 x MFMA) cycles over 1024 SIMDs x 2.4 GHz x duration; `MFMA share` = the MFMAs' part of those cycles. "
          "`K loop` = the innermost loop holding MFMAs: trips = dyn MFMA / static MFMA of that loop, `in` = its share of the dynamic VALU, the static mix of "
          "its VALU follows (per MFMA); `outside` = what is left (prologue: staging, BatchNorm fold of the input; epilogue: statistics partials, stores) with the static mix of the code outside the loop.\n")
    hdr = "| kernel | us/step | dyn VALU/launch | dyn MFMA/launch | pipe | MFMA share | K loop: VALU per MFMA (fma32 / f64 / cvt / cmp-sel / int-addr), LDS, VMEM per MFMA | VALU in / outside K loop | outside mix: fma32 / f64 / cvt / cmp-sel / int-addr |"
    print(hdr)
    print("|" + "---|" * (hdr.count("|") - 1))
    tot_cycles = tot_mfma_cycles = tot_budget = 0.0
    details = []
    for name, nl, avg_us, us_step, bound in rows:
        syms = by_short.get(name) or by_short.get(name.replace(", false", ", 0")) or []
        p = pmc.get(name)
        if not syms or p is None:
            continue
        sym = syms[0]
        ins, L = kern[sym], labs[sym]
        loops, depth = analyse(ins, L)
        stat = Counter(c for _, _, c, _, _ in ins)
        dyn_valu = p["valu_per_launch"] - p["mfma_per_launch"]          # SQ_INSTS_VALU includes the MFMAs
        dyn_mfma = p["mfma_per_launch"]
        mc = max([mfma_cycles(op) for _, op, c, _, _ in ins if c == "mfma"] or [64])
        # fp64 VALU issues at half the fp32 rate (8 cycles per wave instruction; MI355X: 78.6 vs 157.3 TFLOP/s vector): the counters do not split VALU by
        # type, so the static f64 share of the kernel's VALU is applied to the dynamic count
        vs = sum(stat[c] for c in ("fma32", "f64", "cvt/trans", "cmp/sel", "int/addr")) or 1
        f64_share = stat["f64"] / vs
        cycles = 4.0 * dyn_valu * (1.0 + f64_share) + mc * dyn_mfma
        budget = 1024 * 2400.0 * avg_us
        tot_cycles += cycles * nl; tot_mfma_cycles += mc * dyn_mfma * nl; tot_budget += budget * nl
        cell_loop = cell_split = cell_out = "-"
        has_mfma_loop = dyn_mfma > 0 and any(c == "mfma" and dp > 0 for (_, _, c, _, _), dp in zip(ins, depth))
        if dyn_mfma > 0 and not has_mfma_loop:
            # K loop fully unrolled (the chunk loop of k_conv_mfma with PD weight chunks in flight): one pass over the kernel per tile
            vset = ("fma32", "f64", "cvt/trans", "cmp/sel", "int/addr")
            nm = stat["mfma"] or 1
            cell_loop = (f"unrolled flat, per MFMA: {sum(stat[c] for c in vset) / nm:.2f} (" + " / ".join(f"{stat[c] / nm:.2f}" for c in vset) + f"), {stat['lds'] / nm:.2f}, {stat['vmem'] / nm:.2f}")
            passes = dyn_mfma / nm
            cell_split = f"static VALU x passes = {100 * sum(stat[c] for c in vset) * passes / max(dyn_valu, 1):.0f} % of dyn"
            cell_out = "(no loop: prologue, K steps and epilogue are one flat pass)"
        elif dyn_mfma > 0 and loops:
            # innermost loop holding MFMAs = the one with the fewest instructions among the loops that contain an MFMA
            cand = []
            for (lo, hi) in loops:
                body = [(c, op) for addr, op, c, _, _ in ins if lo <= addr <= hi]
                nm = sum(1 for c, _ in body if c == "mfma")
                if nm:
                    cand.append((len(body), lo, hi, nm, body))
            if cand:
                _, lo, hi, nm, body = min(cand)
                # all MFMA-holding loops at the same level (unrolled siblings) share the trips estimate through the total static MFMA count inside loops
                in_loops = [(c, op) for (addr, op, c, _, _), dp in zip(ins, depth) if dp > 0]
                nm_all = sum(1 for c, _ in in_loops if c == "mfma")
                waves = dyn_mfma / max(nm_all, 1)                        # wave-trips through "all loop code once"
                lc = Counter(c for c, _ in in_loops)
                valu_in = sum(lc[c] for c in ("fma32", "f64", "cvt/trans", "cmp/sel", "int/addr")) * waves
                valu_in = min(valu_in, dyn_valu)
                per = lambda c: lc[c] / max(nm_all, 1)
                cell_loop = (f"{sum(lc[c] for c in ('fma32','f64','cvt/trans','cmp/sel','int/addr')) / max(nm_all,1):.2f} "
                             f"({per('fma32'):.2f} / {per('f64'):.2f} / {per('cvt/trans'):.2f} / {per('cmp/sel'):.2f} / {per('int/addr'):.2f}), "
                             f"{per('lds'):.2f}, {per('vmem'):.2f}")
                cell_split = f"{100 * valu_in / max(dyn_valu, 1):.0f} % / {100 * (1 - valu_in / max(dyn_valu, 1)):.0f} %"
                oc = Counter(c for (addr, op, c, _, _), dp in zip(ins, depth) if dp == 0)
                ov = sum(oc[c] for c in ("fma32", "f64", "cvt/trans", "cmp/sel", "int/addr")) or 1
                cell_out = " / ".join(f"{100 * oc[c] / ov:.0f} %" for c in ("fma32", "f64", "cvt/trans", "cmp/sel", "int/addr"))
        else:
            # VALU kernel: the mix of the deepest loop level that holds at least 30 % of the static VALU, else the whole kernel
            vset = ("fma32", "f64", "cvt/trans", "cmp/sel", "int/addr")
            best = None
            for dp in sorted(set(depth), reverse=True):
                c = Counter(cl for (_, _, cl, _, _), d2 in zip(ins, depth) if d2 >= dp)
                if sum(c[x] for x in vset) >= 0.3 * sum(stat[x] for x in vset):
                    best = (dp, c); break
            dp, c = best if best else (0, stat)
            v = sum(c[x] for x in vset) or 1
            cell_loop = f"hot code (loop depth >= {dp}): " + " / ".join(f"{100 * c[x] / v:.0f} %" for x in vset) + f"; LDS {c['lds'] / v:.2f}, VMEM {c['vmem'] / v:.2f} per VALU"
        share = mc * dyn_mfma / cycles if cycles else 0.0
        print(f"| `{name}` | {us_step:.0f} | {dyn_valu / 1e6:.1f} M | {dyn_mfma / 1e6:.2f} M | {100 * cycles / budget:.0f} % | {100 * share:.0f} % | {cell_loop} | {cell_split} | {cell_out} |")
        details.append((name, us_step, cycles * nl, mc * dyn_mfma * nl, budget * nl, stat))
    print(f"\nStep total over these kernels: {tot_budget / (1024 * 2400.0):.0f} us; their instructions account for {100 * tot_cycles / tot_budget:.0f} % of the vector pipe's cycles in that time, "
          f"the MFMAs alone for {100 * tot_mfma_cycles / tot_budget:.0f} %.  If every non-MFMA VALU instruction vanished and nothing ever stalled, the step would take "
          f"{tot_mfma_cycles / (1024 * 2400.0):.0f} us; with the VALU work as it is and no stalls, {tot_cycles / (1024 * 2400.0):.0f} us.\n")
    print("## Where the non-MFMA pipe cycles are (4 cycles per plain VALU instruction), by kernel\n")
    print("| kernel | VALU pipe us/step | MFMA pipe us/step | measured us/step | stall + wait us/step |")
    print("|---|---|---|---|---|")
    for name, us_step, cyc, mcyc, bud, stat in sorted(details, key=lambda d: -(d[2] - d[3])):
        f = 1.0 / (1024 * 2400.0)
        print(f"| `{name}` | {(cyc - mcyc) * f:.0f} | {mcyc * f:.0f} | {bud * f:.0f} | {(bud - cyc) * f:.0f} |")
    print("\n## Static instruction counts (whole kernel)\n")
    print("| kernel | " + " | ".join(CLASSES) + " | loops |")
    print("|---|" + "---|" * (len(CLASSES) + 1))
    for name, nl, avg_us, us_step, bound in rows:
        syms = by_short.get(name) or []
        if not syms:
            continue
        ins, L = kern[syms[0]], labs[syms[0]]
        loops, _ = analyse(ins, L)
        stat = Counter(c for _, _, c, _, _ in ins)
        print(f"| `{name}` | " + " | ".join(str(stat[c]) for c in CLASSES) + f" | {len(loops)} |")
    missing = [n for n, *_ in rows if n not in by_short and n.replace(", false", ", 0") not in by_short]
    if missing:
        print("\nNot matched to a symbol of the library (name spelling): " + ", ".join(f"`{m}`" for m in missing))


if __name__ == "__main__":
    main()
