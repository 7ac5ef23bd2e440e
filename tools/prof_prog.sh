#!/bin/bash
# ONE purpose: one rocprofv3 pass over an arbitrary program of this repository (run ON THE GPU BOX):
#   gpurun -- 'bash tools/prof_prog.sh r03 lat1_sq "--pmc SQ_WAVE_CYCLES SQ_WAIT_ANY" python3 tools/lat_once.py --batch 1'
#   gpurun -- 'bash tools/prof_prog.sh r03 cb_trace "--kernel-trace --stats" python3 tools/bench_configs.py cb --reps 2'
# PROF_TIMEOUT (seconds, default 600) bounds the pass.  TCC has 4 counter slots per pass: FETCH_SIZE takes 3, WRITE_SIZE 2 -- one of them per pass
# (a request that does not fit aborts inside the first dispatch and the process then hangs until the timeout).
# $3 = the profiler's mode flags (counters only, or trace only: never both); the program itself follows directly
# after `--` (no shell hop between the profiler and the program).  Prints per-kernel counter means.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
R=$1; TAG=$2; MODE=$3; shift 3
OUT=gpurun_out/prof_$R
mkdir -p "$OUT"
# shellcheck disable=SC2086
timeout "${PROF_TIMEOUT:-600}" rocprofv3 $MODE --output-format csv -d "$OUT/$TAG" -- "$@" > "$OUT/$TAG.out" 2> "$OUT/$TAG.err"
echo "rc=$?"
find "$OUT/$TAG" -name '*.db' -delete 2>/dev/null
python3 - "$OUT/$TAG" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        a = agg[r["Kernel_Name"][:70]][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
for k, cs in agg.items():
    print(k)
    for c, (n, t) in sorted(cs.items()):
        print("   %-30s n=%d per-dispatch %.5g" % (c, n, t / n))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    print(open(f).read())
PY
tail -3 "$OUT/$TAG.out"
