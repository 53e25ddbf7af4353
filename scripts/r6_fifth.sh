#!/bin/bash
mkdir -p gpurun_out
{
echo "== main library: scattered patch order and suspension"
export REPS=3
export SCHED_CONFIGS='[{"PBRHIP_PATCH_SHUFFLE":"0","PBRHIP_SUSP_TURNS":"0"},{"PBRHIP_PATCH_SHUFFLE":"1","PBRHIP_SUSP_TURNS":"0"},{"PBRHIP_PATCH_SHUFFLE":"1","PBRHIP_SUSP_TURNS":"8"},{"PBRHIP_PATCH_SHUFFLE":"1","PBRHIP_SUSP_TURNS":"16"},{"PBRHIP_PATCH_SHUFFLE":"0","PBRHIP_SUSP_TURNS":"8"}]'
timeout 900 python scripts/sched_ab.py ggx 2>&1 | grep -v "^sched\|amdgpu.ids\|RCCL\|HIP version\|ROCm\|Hostname\|Librccl"
export SCHED_CONFIGS='[{"PBRHIP_PATCH_SHUFFLE":"1","PBRHIP_SUSP_TURNS":"8"},{"PBRHIP_PATCH_SHUFFLE":"0","PBRHIP_SUSP_TURNS":"8"}]'
for lib in build/g4/libpbrhip.so build/b256/libpbrhip.so build/g4b256/libpbrhip.so; do
  echo "== $lib"
  PBRHIP_LIB=$(realpath $lib) timeout 600 python scripts/sched_ab.py ggx 2>&1 | grep "^{"
done
echo "== wave log, eighth of C2, scattered patches, PBRHIP_SUSP_TURNS=8"
PBRHIP_SUSP_TURNS=8 timeout 300 python scripts/wave_log.py 8 2>&1 | grep launch
echo "== the same, library g4b256"
PBRHIP_LIB=$(realpath build/g4b256/libpbrhip.so) PBRHIP_SUSP_TURNS=8 timeout 300 python scripts/wave_log.py 8 2>&1 | grep launch
} > gpurun_out/r6_fifth.txt 2>&1
cat gpurun_out/r6_fifth.txt
