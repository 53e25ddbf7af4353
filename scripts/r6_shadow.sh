#!/bin/bash
mkdir -p gpurun_out
{
echo "== parity (whole file): shadow rays suspended too (scenes without media)"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -5
export REPS=3
export SCHED_CONFIGS='[{"PBRHIP_SUSP_TURNS":"0"},{"PBRHIP_SUSP_TURNS":"8"},{"PBRHIP_SUSP_TURNS":"16"},{"PBRHIP_SUSP_TURNS":"24"},{"PBRHIP_SUSP_TURNS":"32"},{"PBRHIP_SUSP_TURNS":"0"}]'
timeout 900 python scripts/sched_ab.py ggx 2>&1 | grep "^{\|^1/8\|^!!"
echo "== wave log, eighth of C2, PBRHIP_SUSP_TURNS=8"
PBRHIP_SUSP_TURNS=8 timeout 300 python scripts/wave_log.py 8 2>&1 | grep launch
echo "== hair frame, suspension 0 / 16"
VARIANT=hair SPP=128 REPS=2 timeout 600 python scripts/frame_ab.py "PBRHIP_SUSP_TURNS=0" "PBRHIP_SUSP_TURNS=16" "PBRHIP_SUSP_TURNS=24" 2>&1 | grep "ms$"
} > gpurun_out/r6_shadow.txt 2>&1
cat gpurun_out/r6_shadow.txt
