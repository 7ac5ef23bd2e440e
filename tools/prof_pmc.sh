#!/bin/bash
# ONE purpose: one rocprofv3 --pmc pass (counters only, no tracing) of a 1-step bench (run ON THE GPU BOX):
#   gpurun --timeout 400 -- 'bash tools/prof_pmc.sh r02 sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY ...'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
R=$1; TAG=$2; shift 2
OUT=gpurun_out/prof_$R
mkdir -p "$OUT"
# shellcheck disable=SC2068
timeout 300 rocprofv3 --pmc $@ --output-format csv -d "$OUT/pmc_$TAG" -- \
    python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --headline-only ${BENCH_ARGS:-} > "$OUT/pmc_$TAG.json" 2> "$OUT/pmc_$TAG.err"
echo "rc=$?"
find "$OUT/pmc_$TAG" -name '*.db' -delete 2>/dev/null
python3 - "$OUT/pmc_$TAG" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        a = agg[r["Kernel_Name"][:60]][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
for k, cs in agg.items():
    if "blind_rotate" in k or "keyswitch" in k:
        print(k)
        for c, (n, t) in cs.items():
            print("   %-28s n=%d per-dispatch %.4g" % (c, n, t / n))
PY
