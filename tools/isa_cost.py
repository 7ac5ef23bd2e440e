#!/usr/bin/env python3
"""Static instruction mix of the blind-rotation kernel's CMux loop, from the gfx950 ISA hipcc emits.

    python tools/isa_cost.py [--asm FILE] [--generic]

Compiles csrc/tfhe_amd.hip to assembly (device only), finds the gate-set k_blind_rotate instantiation
(N=1024, 8 waves, digits in pairs, l = 2 and Bgbit = 10 at compile time: the two transform groups of a CMux
are unrolled, so the CMux loop is one depth-1 loop) and counts instructions per pipe in it.  Blocks reached only
through the exact-rounding fallback or the non-rotating path are listed separately.
Cycle weights: the issue costs MEASURED on MI355X (profiles/r02_ubench_issue.txt): fp64 VALU 4.3 cycles per
wave64 instruction, 32-bit VALU ~3.3 (2.3 for add/xor/sub, 4.4 for bfe/shift/three-operand forms), LDS
per-instruction cycles of MI355X_MICROARCH.md (confirmed by the same microbenchmark).
A static model to read PMC counts against (profiles/r02_final_pmc.txt), not a measurement."""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# instantiations with the gadget length fixed at compile time (LC = 2: the gate set); --generic: LC = 0
KERNELS = {0: "k_blind_rotateIiLi10ELi8ELi2ELi%dELi%dE"}
LDS_CYCLES = {"ds_add_u32": 4, "ds_read_b32": 2, "ds_read_b64": 2, "ds_read_b128": 4, "ds_read2_b32": 4, "ds_read2_b64": 8, "ds_read2st64_b32": 4,
              "ds_read2st64_b64": 8, "ds_write_b32": 4, "ds_write_b64": 6, "ds_write_b128": 13, "ds_write2_b32": 6,
              "ds_write2st64_b32": 6, "ds_write2_b64": 13, "ds_write2st64_b64": 13}


def classify(op):
    if op.startswith("v_") and "f64" in op:
        return "fp64"
    if op.startswith("v_"):
        return "valu32"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    return "salu"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm", default=None, help="existing assembly file (skip compilation)")
    ap.add_argument("--generic", action="store_true", help="the instantiation that reads the gadget length at run time")
    a = ap.parse_args()
    asm = a.asm
    if asm is None:
        asm = os.path.join(tempfile.mkdtemp(prefix="isa_cost_"), "tfhe_amd.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-S",
                               "--cuda-device-only", "-o", asm, os.path.join(ROOT, "experimental-tfhe_amd", "csrc", "tfhe_amd.hip")],
                              stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
    a.variant = 0
    key = KERNELS[0] % ((0, 0) if a.generic else (2, 10))
    start = next(i for i, ln in enumerate(lines) if ln.startswith("_ZN4tfhe14" + key) and ":" in ln)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    # split into blocks with their loop depth
    blocks, cur = [], {"label": "entry", "depth": 0, "ins": []}
    for ln in lines[start + 1:end]:
        t = ln.strip()
        m = re.match(r"^(\.LBB\d+_\d+):|^; %bb\.(\d+):", t)
        if m:
            blocks.append(cur)
            d = re.search(r"Depth[= ](\d+)", ln)
            cur = {"label": m.group(1) or "bb." + m.group(2), "depth": int(d.group(1)) if d else 0, "ins": []}
            continue
        if not t or t.startswith((";", ".")):
            continue
        cur["ins"].append(t)
    blocks.append(cur)
    # the exact-rounding fallback: blocks that use v_floor_f64 / v_ldexp_f64 (only that path does); the
    # non-rotating path (plain external product): blocks that read the accumulator without v_and_or_b32 but
    # sit between the loop head and the first transform -- recognised by ds_read2st64_b32 without v_xad_u32
    for b in blocks:
        ops = [i.split()[0] for i in b["ins"]]
        b["cold"] = any(o.startswith(("v_floor_f64", "v_ldexp_f64")) for o in ops) or (
            any(o.startswith("ds_read2st64_b32") for o in ops) and not any(o.startswith("v_xad_u32") for o in ops)
            and not any("f64" in o for o in ops))
    inner_trips = 2 if a.generic else 1  # generic: the digit-group loop is a depth-2 loop run (k+1) times
    hot = collections.Counter()
    lds_ops = collections.Counter()
    lds_cycles = 0.0
    cold = collections.Counter()
    for b in blocks:
        if b["depth"] == 0:
            continue
        w = inner_trips if b["depth"] >= 2 else 1
        for i in b["ins"]:
            op = i.split()[0]
            k = classify(op)
            if b["cold"]:
                cold[k] += w
                continue
            hot[k] += w
            if k == "lds":
                base = op.replace("_e32", "").replace("_e64", "")
                lds_cycles += w * LDS_CYCLES.get(base, 4)
                lds_ops[base] += w
    waves_per_simd = 2
    valu32_cost = 3.3
    simd = hot["fp64"] * 4.3 + hot["valu32"] * valu32_cost
    print(f"kernel {key} (variant {a.variant}), per CMux per wave, hot path:")
    for k in ("fp64", "valu32", "lds", "vmem", "salu", "branch", "wait", "scratch"):
        print(f"  {k:8s} {hot[k]:6d}")
    print(f"  fallback-only blocks (exact rounding): {dict(cold)}")
    print(f"SIMD issue cycles per CMux per wave : {simd:.0f}  (fp64 x4.3 + valu32 x{valu32_cost})")
    print(f"LDS cycles per CMux per wave        : {lds_cycles:.0f}")
    print("  " + ", ".join(f"{n} x{c} ({c * LDS_CYCLES.get(n, 4)} cyc)" for n, c in sorted(lds_ops.items())))
    wpc = 4 * waves_per_simd
    cu_cycles = max(simd * waves_per_simd, lds_cycles * wpc) / wpc
    print(f"{wpc} waves per CU: SIMD {simd * waves_per_simd:.0f} cycles vs LDS {lds_cycles * wpc:.0f} cycles per round of {wpc} CMux")
    for ghz in (2.4, 2.05):
        print(f"  ceiling at {ghz} GHz, perfect overlap: {256 * ghz * 1e9 / cu_cycles / 630 / 1e3:.0f} k bootstraps/s; "
              f"no overlap: {256 * ghz * 1e9 / ((simd * waves_per_simd + lds_cycles * wpc) / wpc) / 630 / 1e3:.0f} k")


if __name__ == "__main__":
    sys.exit(main())
