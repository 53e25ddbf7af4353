#!/bin/bash
mkdir -p gpurun_out
{
for f in 0 1 2 3 5 7; do
echo "== PBRHIP_SSS_FOREIGN=$f"
PBRHIP_SSS_FOREIGN=$f PBRHIP_DEBUG=1 PBRHIP_PV_STATS=1 VARIANT=sss SPP=8 timeout 600 python scripts/qtree_probe.py 2>&1 | grep "WIDE=1\|^walk\|random walks" | head -6
PBRHIP_SSS_FOREIGN=$f VARIANT=sss SPP=256 REPS=1 timeout 600 python scripts/frame_ab.py "" 2>&1 | grep "ms$"
done
} > gpurun_out/r6_sss2.txt 2>&1
cat gpurun_out/r6_sss2.txt
