#!/bin/bash
mkdir -p gpurun_out
{
echo "== hair / curve parity with the two-piece curve turn (main lib)"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "curve or hair or soup or trace_hooks" 2>&1 | tail -4
echo "== suspension counts (eighth) and frame times"
export REPS=2
export SCHED_CONFIGS='[{"PBRHIP_SUSP_TURNS":"0"},{"PBRHIP_SUSP_TURNS":"8"},{"PBRHIP_SUSP_TURNS":"24"},{"PBRHIP_SUSP_TURNS":"64"}]'
timeout 600 python scripts/sched_ab.py ggx 2>&1 | grep -v "^sched\|amdgpu.ids\|RCCL\|HIP version\|ROCm\|Hostname\|Librccl"
echo "== wave log, eighth of C2, no suspension"
STATS=1 PBRHIP_SUSP_TURNS=0 timeout 300 python scripts/wave_log.py 8 2>&1 | grep launch
echo "== wave log, eighth of C2, suspension after 16 turns"
STATS=1 PBRHIP_SUSP_TURNS=16 timeout 300 python scripts/wave_log.py 8 2>&1 | grep launch
echo "== hair frame (128 spp), library variants"
for lib in pbrlab_amd/libpbrhip.so build/p0/libpbrhip.so build/p1b5/libpbrhip.so build/p0b5/libpbrhip.so; do
  echo "-- $lib"
  PBRHIP_LIB=$(realpath $lib) VARIANT=hair SPP=128 REPS=2 timeout 600 python scripts/frame_ab.py "" 2>&1 | grep "ms$"
done
} > gpurun_out/r6_second.txt 2>&1
cat gpurun_out/r6_second.txt
