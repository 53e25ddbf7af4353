#!/bin/bash
mkdir -p gpurun_out
{
for t in 0 8; do
echo "== wave log (no statistics build), eighth of C2, PBRHIP_SUSP_TURNS=$t"
PBRHIP_SUSP_TURNS=$t timeout 300 python scripts/wave_log.py 8 2>&1 | grep launch
done
echo "== wave log, whole C2 frame as one group, PBRHIP_SUSP_TURNS=0"
PBRHIP_SUSP_TURNS=0 timeout 300 python scripts/wave_log.py 1 2>&1 | grep launch
echo "== wave log, whole C2 frame as one group, PBRHIP_SUSP_TURNS=8"
PBRHIP_SUSP_TURNS=8 timeout 300 python scripts/wave_log.py 1 2>&1 | grep launch
} > gpurun_out/r6_fourth.txt 2>&1
cat gpurun_out/r6_fourth.txt
