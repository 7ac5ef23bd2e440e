#!/bin/bash
# ONE purpose: rocprofv3 kernel trace + stats of the default bench command (run ON THE GPU BOX):
#   gpurun --timeout 400 -- 'bash tools/prof_trace.sh r02'
# The program itself follows `--` (no shell hop) and runs with --no-cpu-baseline (no children
# from a process the profiler has already GPU-initialised).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
R=${1:-r02}; shift
OUT=gpurun_out/prof_$R
mkdir -p "$OUT"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- \
    python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --headline-only "$@" > "$OUT/bench_traced.json" 2> "$OUT/trace.err"
echo "rc=$?"
find "$OUT/trace" -name '*.db' -delete 2>/dev/null
find "$OUT/trace" -name '*kernel_stats.csv' | head -1 | xargs -r head -12
tail -c 400 "$OUT/bench_traced.json"
