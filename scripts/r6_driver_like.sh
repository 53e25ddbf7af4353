#!/bin/bash
mkdir -p gpurun_out
{
t0=$(date +%s)
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
t1=$(date +%s); echo "smoke: $((t1-t0)) s"
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "rc $?"; t2=$(date +%s); echo "bench default: $((t2-t1)) s"
wc -l gpurun_out/bench_default.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","higher_is_better","scaling","vs_baseline","dtype","data","config")})
print("roofline", {k:d["roofline"].get(k) for k in ("bound","achieved","peak","unit","frac","traffic")})
print("cpu_baseline", {k:d["cpu_baseline"].get(k) for k in ("value","unit","cores","kind")})
print("host_layer", d["host_layer"]["value"], d["host_layer"]["ms_per_step"])
PY
tail -3 gpurun_out/bench_default.err
} > gpurun_out/r6_driver_like.txt 2>&1
cat gpurun_out/r6_driver_like.txt
