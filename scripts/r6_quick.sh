#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "render_matches or resumable or patch_order or random_materials or trace_hooks or concurrent or group_schedules or sharding or cancel or chunking or render_multi or instance" 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -6 > gpurun_out/r6_quick.txt
cat gpurun_out/r6_quick.txt
