for v in 0 1 2; do timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --br-variant $v > gpurun_out/r02_var_br$v.json 2>&1; done
timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --lib experimental-tfhe_amd/libtfhe_amd_asmlds.so > gpurun_out/r02_var_asmlds.json 2>&1
timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --ks-stream > gpurun_out/r02_var_ksstream.json 2>&1
timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --ks-gather > gpurun_out/r02_var_ksgather.json 2>&1
timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --streamed > gpurun_out/r02_var_streamed.json 2>&1
for b in 512 1024 2048 8192 16384; do timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --batch $b > gpurun_out/r02_var_batch$b.json 2>&1; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02_var_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["value"]), d["kernels_ms"], d.get("streamed_schedule"))
    except Exception as e:
        print(f, "ERR", open(f).read()[-300:])
PY
