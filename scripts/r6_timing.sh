#!/bin/bash
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider -k "timing_flags or without_classify" 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tail -3
for i in 1 2; do
python bench.py --no-cpu-baseline --no-live-pmc --steps 5 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('trace-only events:', round(d['value'],1), round(d['ms_per_step'],2), 'host_layer', round(d['host_layer']['ms_per_step'],2), 'avg_launch_ms', round(d['roofline']['avg_launch_ms'],3), 'frac', round(d['roofline']['frac'],3))"
python bench.py --no-cpu-baseline --no-live-pmc --no-roofline --steps 5 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no events:        ', round(d['value'],1), round(d['ms_per_step'],2), 'host_layer', round(d['host_layer']['ms_per_step'],2))"
done
} > gpurun_out/r6_timing.txt 2>&1
cat gpurun_out/r6_timing.txt
