#!/bin/bash
mkdir -p gpurun_out
{
echo "== hair frame (128 spp): vote weight of the curve phase, refill threshold"
for lib in pbrlab_amd/libpbrhip.so build/wc1/libpbrhip.so build/wc3/libpbrhip.so build/rc16/libpbrhip.so build/rc32/libpbrhip.so pbrlab_amd/libpbrhip.so; do
  echo "-- $lib"
  PBRHIP_LIB=$(realpath $lib) VARIANT=hair SPP=128 REPS=2 timeout 600 python scripts/frame_ab.py "" 2>&1 | grep "ms$"
done
echo "== C2 frame (64 spp): the top of the Q tree staged in LDS (1 / 16 / 64 nodes) vs not"
for lib in pbrlab_amd/libpbrhip.so build/top1/libpbrhip.so build/top16/libpbrhip.so build/top64/libpbrhip.so pbrlab_amd/libpbrhip.so; do
  echo "-- $lib"
  PBRHIP_LIB=$(realpath $lib) VARIANT=ggx SPP=64 REPS=2 timeout 600 python scripts/frame_ab.py "" "PBRHIP_STREAMS=1" 2>&1 | grep "ms$"
done
} > gpurun_out/r6_tune.txt 2>&1
cat gpurun_out/r6_tune.txt
