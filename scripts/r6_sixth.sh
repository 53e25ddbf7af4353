#!/bin/bash
mkdir -p gpurun_out
{
echo "== parity (whole file), block pools + scattered patches + suspension after 24 turns (defaults)"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
export REPS=3
export SCHED_CONFIGS='[{"PBRHIP_SUSP_TURNS":"0"},{"PBRHIP_SUSP_TURNS":"8"},{"PBRHIP_SUSP_TURNS":"16"},{"PBRHIP_SUSP_TURNS":"32"}]'
for lib in pbrlab_amd/libpbrhip.so build/nopool/libpbrhip.so pbrlab_amd/libpbrhip.so; do
  echo "== $lib"
  PBRHIP_LIB=$(realpath $lib) timeout 600 python scripts/sched_ab.py ggx 2>&1 | grep "^{\|^1/8"
done
echo "== wave log, eighth of C2, block pools, PBRHIP_SUSP_TURNS=8"
PBRHIP_SUSP_TURNS=8 timeout 300 python scripts/wave_log.py 8 2>&1 | grep launch
echo "== wave log, whole frame as one group, block pools, PBRHIP_SUSP_TURNS=8"
PBRHIP_SUSP_TURNS=8 timeout 300 python scripts/wave_log.py 1 2>&1 | grep launch
} > gpurun_out/r6_sixth.txt 2>&1
cat gpurun_out/r6_sixth.txt
