#!/bin/bash
mkdir -p gpurun_out
{
export REPS=3 SCHED_CONFIGS='[{}]'
for round in 1 2; do
for lib in build/h1/libpbrhip.so pbrlab_amd/libpbrhip.so build/hb256/libpbrhip.so; do
  echo "== $lib"
  PBRHIP_LIB=$(realpath $lib) timeout 600 python scripts/sched_ab.py ggx 2>&1 | grep "^{"
  PBRHIP_LIB=$(realpath $lib) VARIANT=hair SPP=128 REPS=1 timeout 600 python scripts/frame_ab.py "" 2>&1 | grep "ms$"
  PBRHIP_LIB=$(realpath $lib) VARIANT=sss SPP=256 REPS=1 timeout 600 python scripts/frame_ab.py "" 2>&1 | grep "ms$"
done
done
} > gpurun_out/r6_heads2.txt 2>&1
cat gpurun_out/r6_heads2.txt
