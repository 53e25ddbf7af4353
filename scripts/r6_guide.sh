#!/bin/bash
mkdir -p gpurun_out
{
export SCHED_CONFIGS='[{}]'
for lib in pbrlab_amd/libpbrhip.so build/g1/libpbrhip.so build/g3/libpbrhip.so build/g4/libpbrhip.so build/g6/libpbrhip.so pbrlab_amd/libpbrhip.so build/g3/libpbrhip.so build/g4/libpbrhip.so; do
  echo "== $lib"
  PBRHIP_LIB=$(realpath $lib) REPS=5 python scripts/sched_ab.py ggx 2>&1 | grep "world1\|!!" | cut -c90-330
done
} > gpurun_out/r6_guide.txt 2>&1
cat gpurun_out/r6_guide.txt
