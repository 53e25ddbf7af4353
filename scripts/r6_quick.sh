#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "instance_transforms or resumable or patch_order or render_matches or random_materials or degenerate or doomed" 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -15 > gpurun_out/r6_quick.txt
cat gpurun_out/r6_quick.txt
