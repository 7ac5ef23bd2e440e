#!/bin/bash
# Run ON THE GPU BOX (through gpurun) to produce the bench line and the rocprofv3 evidence of a
# round:   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r02'
# Outputs land in gpurun_out/prof_<round>/ (merged back by gpurun); copy the summaries you want
# judged into profiles/ with   python tools/summarize_prof.py gpurun_out/prof_<round> profiles/<round>
#
# Rules this script follows (see the task's GPU-pool notes):
#   * rocprofv3 gets the program itself after `--` (python3 bench.py ...), never a shell hop;
#   * under rocprofv3 bench.py runs with --no-cpu-baseline: the profiler initialises the GPU before
#     the program starts, so the program must not spawn children;
#   * --pmc passes are separate from --kernel-trace/--stats passes, and FETCH_SIZE / WRITE_SIZE
#     each get their own pass (TCC slot limits).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
R=${1:-r02}
OUT=gpurun_out/prof_$R
mkdir -p "$OUT"
VARIANTS=${VARIANTS:-"0 1 2"}

echo "== plain bench (default schedule, with CPU baseline)" | tee "$OUT/log.txt"
timeout 600 python3 bench.py --steps 5 --warmup 1 --streamed > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 600 "$OUT/bench.json" | tee -a "$OUT/log.txt"

for v in $VARIANTS; do
  echo "== bench, blind-rotation variant $v" | tee -a "$OUT/log.txt"
  timeout 300 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --br-variant "$v" \
      > "$OUT/bench_variant$v.json" 2> "$OUT/bench_variant$v.err"
  python3 - "$OUT/bench_variant$v.json" <<'PY' | tee -a "$OUT/log.txt"
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("   value %.0f bootstraps/s  ms/step %.2f  kernels %s  fp64 frac %.3f" % (
        d["value"], d["ms_per_step"], d["kernels_ms"], d["roofline"]["fp64_valu"]["frac"]))
except Exception as e:
    print("   (no result:", e, ")")
PY
done

echo "== bench, key switch by the streaming kernel" | tee -a "$OUT/log.txt"
timeout 300 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --ks-stream > "$OUT/bench_ksstream.json" 2> "$OUT/bench_ksstream.err"
tail -c 300 "$OUT/bench_ksstream.json" | tee -a "$OUT/log.txt"

echo "== experiment build: hand-placed ds_read_b64 in the transposes (parity first, then timing)" | tee -a "$OUT/log.txt"
TFHE_AMD_TEST_LIB=experimental-tfhe_amd/libtfhe_amd_asmlds.so timeout 300 python3 tests/gpu_stage_check.py 4 2>&1 | tail -4 | tee -a "$OUT/log.txt"
timeout 300 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --lib experimental-tfhe_amd/libtfhe_amd_asmlds.so \
    > "$OUT/bench_asmlds.json" 2> "$OUT/bench_asmlds.err"
tail -c 400 "$OUT/bench_asmlds.json" | tee -a "$OUT/log.txt"

echo "== ablation table (diagnostic build, wrong results by design, times only)" | tee -a "$OUT/log.txt"
timeout 600 python3 tools/ablate.py --reps 3 2> "$OUT/ablate.err" | tee "$OUT/ablate.txt" | tee -a "$OUT/log.txt"

echo "== BASELINE configs 3 and 4 (circuit bootstrap + LUT evaluation, batched N=2048 transforms)" | tee -a "$OUT/log.txt"
timeout 900 python3 tools/bench_configs.py all > "$OUT/configs.jsonl" 2> "$OUT/configs.err"
cut -c1-300 "$OUT/configs.jsonl" | tee -a "$OUT/log.txt"
for w in 8 12; do
  timeout 300 python3 tools/bench_configs.py fft --fft-waves $w > "$OUT/configs_fftwaves$w.jsonl" 2>> "$OUT/configs.err"
  cut -c1-200 "$OUT/configs_fftwaves$w.jsonl" | tee -a "$OUT/log.txt"
done

echo "== rocprofv3 kernel trace + stats" | tee -a "$OUT/log.txt"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- \
    python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_traced.json" 2> "$OUT/trace.err"

for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE"; do
  tag=$(echo "$grp" | tr ' ' '_' | cut -c1-40)
  echo "== rocprofv3 --pmc $grp" | tee -a "$OUT/log.txt"
  # shellcheck disable=SC2086
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$tag" -- \
      python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_$tag.json" 2> "$OUT/pmc_$tag.err"
done
# keep the merge-back small: drop per-dispatch traces that are not summaries
find "$OUT" -name '*.db' -delete 2>/dev/null
du -sh "$OUT" | tee -a "$OUT/log.txt"
