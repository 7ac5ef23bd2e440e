#!/usr/bin/env python3
"""Condense the rocprofv3 CSV output of tools/profile_round.sh into small text files for profiles/.

    python tools/summarize_prof.py gpurun_out/prof_r02 profiles/r02

Writes <prefix>_kernel_stats.txt (per-kernel calls / total / average duration from the
kernel-trace pass) and <prefix>_pmc.txt (per-kernel counter sums per dispatch, with the gfx950
FETCH_SIZE x2 correction of MI355X_MICROARCH.md applied and stated), plus copies of the bench lines."""
import csv
import glob
import os
import shutil
import sys
from collections import defaultdict


def find(root, pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


def kernel_stats(root, out):
    rows = []
    for f in find(os.path.join(root, "trace"), "*kernel_trace.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append(r)
    agg = defaultdict(lambda: [0, 0.0])
    for r in rows:
        name = r.get("Kernel_Name") or r.get("kernel_name") or "?"
        try:
            dur = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3  # us
        except Exception:
            continue
        agg[name][0] += 1
        agg[name][1] += dur
    with open(out, "w") as o:
        o.write("kernel-trace summary (rocprofv3 --kernel-trace --stats): calls, total us, average us\n")
        for name, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            o.write(f"{n:8d} {tot:14.1f} {tot / max(n, 1):12.2f}  {name[:150]}\n")
    return bool(rows)


def pmc(root, out):
    with open(out, "w") as o:
        o.write("PMC passes (one rocprofv3 --pmc run each); values are sums over all dispatches of the kernel\n"
                "divided by the dispatch count.  FETCH_SIZE is reported by rocprofv3 in KiB and on gfx950 counts\n"
                "128-B requests as 64 B for wide coalesced reads: the 'corrected' column doubles it.\n\n")
        for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
            if not os.path.isdir(d):
                continue
            agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
            for f in find(d, "*counter_collection.csv"):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        k = r.get("Kernel_Name", "?")
                        c = r.get("Counter_Name", "?")
                        try:
                            v = float(r.get("Counter_Value", "nan"))
                        except ValueError:
                            continue
                        agg[k][c][0] += 1
                        agg[k][c][1] += v
            o.write(f"## {os.path.basename(d)}\n")
            for k, cs in agg.items():
                o.write(f"  {k[:140]}\n")
                for c, (n, tot) in cs.items():
                    per = tot / max(n, 1)
                    extra = f"   corrected bytes/dispatch = {per * 1024 * 2:.0f}" if c == "FETCH_SIZE" else (
                        f"   bytes/dispatch = {per * 1024:.0f}" if c == "WRITE_SIZE" else "")
                    o.write(f"      {c:28s} dispatches {n:6d}  per dispatch {per:16.1f}{extra}\n")
            o.write("\n")


def traffic(root, out):
    """HBM bytes per launch of the blind-rotation kernel from the FETCH_SIZE and WRITE_SIZE passes (KiB units,
    FETCH_SIZE doubled on gfx950 as MI355X_MICROARCH.md prescribes) -> profiles/traffic.json, which bench.py
    reports as roofline.traffic (with this file named as the source)"""
    import json
    per = {}
    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
        n, tot = 0, 0.0
        for d in glob.glob(os.path.join(root, "pmc_" + cname + "*")):
            for f in find(d, "*counter_collection.csv"):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if "k_blind_rotate" in r.get("Kernel_Name", "") and r.get("Counter_Name") == cname:
                            try:
                                tot += float(r["Counter_Value"])
                                n += 1
                            except (KeyError, ValueError):
                                pass
        if n:
            per[cname] = tot / n * 1024 * (2 if cname == "FETCH_SIZE" else 1)
    if len(per) == 2:
        with open(out, "w") as o:
            json.dump({"kernel": "k_blind_rotate", "fetch_bytes_per_launch": per["FETCH_SIZE"],
                       "write_bytes_per_launch": per["WRITE_SIZE"], "bytes_per_launch": per["FETCH_SIZE"] + per["WRITE_SIZE"],
                       "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes under {root} (bench.py --steps 1 --warmup 0, "
                                 "default batch), FETCH_SIZE x2 gfx950 correction applied"}, o, indent=1)
        return True
    return False


def main():
    root, prefix = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
    kernel_stats(root, prefix + "_kernel_stats.txt")
    pmc(root, prefix + "_pmc.txt")
    if traffic(root, os.path.join(os.path.dirname(prefix) or ".", "traffic.json")):
        print("wrote", os.path.join(os.path.dirname(prefix) or ".", "traffic.json"))
    for f in (glob.glob(os.path.join(root, "bench*.json")) + glob.glob(os.path.join(root, "configs*.jsonl")) +
              [os.path.join(root, "log.txt"), os.path.join(root, "ablate.txt")]):
        if os.path.exists(f):
            shutil.copy(f, prefix + "_" + os.path.basename(f))
    print("wrote", prefix + "_*")


if __name__ == "__main__":
    main()
