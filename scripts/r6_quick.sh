#!/bin/bash
mkdir -p gpurun_out
PBRHIP_DEBUG=1 timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "random_walks_start" 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | grep "random walks\|passed\|failed\|Error\|assert" | sort | uniq -c | tail -25 > gpurun_out/r6_quick.txt
cat gpurun_out/r6_quick.txt
