#!/bin/bash
mkdir -p gpurun_out
{
echo "== parity (whole file) with the result-word re-queue"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4
echo "== suspension (eighth) and frame times"
export REPS=3
export SCHED_CONFIGS='[{"PBRHIP_SUSP_TURNS":"0"},{"PBRHIP_SUSP_TURNS":"4"},{"PBRHIP_SUSP_TURNS":"8"},{"PBRHIP_SUSP_TURNS":"16"},{"PBRHIP_SUSP_TURNS":"32"},{"PBRHIP_SUSP_TURNS":"64"},{"PBRHIP_SUSP_TURNS":"0"}]'
timeout 900 python scripts/sched_ab.py ggx 2>&1 | grep -v "^sched\|amdgpu.ids\|RCCL\|HIP version\|ROCm\|Hostname\|Librccl"
echo "== hair: per-ray statistics, pairs off / on (5 blocks per CU both)"
for lib in build/p0b5/libpbrhip.so build/p1b5/libpbrhip.so; do
  echo "-- $lib"
  PBRHIP_LIB=$(realpath $lib) PBRHIP_PV_STATS=1 VARIANT=hair SPP=8 timeout 600 python scripts/qtree_probe.py 2>&1 | grep -v "amdgpu.ids" | grep "WIDE=1\|closest\|shadow\|pv \|curve leaves" | head -12
done
} > gpurun_out/r6_third.txt 2>&1
cat gpurun_out/r6_third.txt
