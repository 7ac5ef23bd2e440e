#!/usr/bin/env python3
"""profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/prof_pmc.sh:

    python tools/make_traffic.py gpurun_out/prof_r02

HBM-side bytes per launch of the blind-rotation kernel (rocprofv3 reports KiB; FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950), tagged with the hash of the kernel
sources it was measured on -- bench.py reports it as roofline.traffic only while that hash matches."""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_sources_sha256)


def main():
    root = sys.argv[1]
    per = {}
    for tag, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        vals = []
        for f in glob.glob(os.path.join(root, "pmc_" + tag, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_blind_rotate" in r["Kernel_Name"] and r["Counter_Name"] == cname:
                    vals.append(float(r["Counter_Value"]))
        if not vals:
            raise SystemExit(f"no {cname} rows under {root}/pmc_{tag}")
        per[cname] = sum(vals) / len(vals) * 1024 * (2 if cname == "FETCH_SIZE" else 1)
    try:
        commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        commit = None
    out = {"kernel": "k_blind_rotate", "fetch_bytes_per_launch": per["FETCH_SIZE"], "write_bytes_per_launch": per["WRITE_SIZE"],
           "bytes_per_launch": per["FETCH_SIZE"] + per["WRITE_SIZE"], "kernel_sources_sha256": bench.kernel_sources_sha256(),
           "measured_at_commit": commit,
           "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/prof_pmc.sh) of bench.py --steps 1 --warmup 0, "
                     f"batch 4096; FETCH_SIZE x2 (gfx950 wide-read correction); commit {commit}"}
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
