#!/bin/bash
mkdir -p gpurun_out
{
echo "== parity: SSS scenes with walk entries (scalar-loaded records)"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "sss or render_matches or random_materials or material_update or instance or render_multi or odd_size" 2>&1 | grep -E "passed|failed|error|Error|FAILED" | tail -8
echo "== C3 frame (256 spp): entries off / on / off / on"
for e in 0 1 0 1; do PBRHIP_SSS_ENTRY=$e VARIANT=sss SPP=256 REPS=1 timeout 600 python scripts/frame_ab.py "" 2>&1 | grep "ms$"; done
echo "== per-kernel, 8 spp"
for e in 1 0; do PBRHIP_SSS_ENTRY=$e PBRHIP_PV_STATS=1 VARIANT=sss SPP=8 timeout 600 python scripts/qtree_probe.py 2>&1 | grep "WIDE=1\|^walk" | head -2; done
echo "== C5 (3840x2160, 16 spp): entries on / off"
for e in 1 0; do PBRHIP_SSS_ENTRY=$e VARIANT=c5 SPP=16 timeout 900 python scripts/qtree_probe.py 2>&1 | grep "WIDE=1"; done
} > gpurun_out/r6_sss.txt 2>&1
cat gpurun_out/r6_sss.txt
