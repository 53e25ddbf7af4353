#!/bin/bash
mkdir -p gpurun_out
{
echo "== C3 frame (256 spp): k_sss_walk refill threshold (scatter when that many lanes are ready; 56 = main) and scatterings per launch (24 = main)"
for lib in pbrlab_amd/libpbrhip.so build/wr32/libpbrhip.so build/wr40/libpbrhip.so build/wr48/libpbrhip.so build/wr62/libpbrhip.so build/wc16/libpbrhip.so build/wc32/libpbrhip.so pbrlab_amd/libpbrhip.so; do
  echo "-- $lib"
  PBRHIP_LIB=$(realpath $lib) VARIANT=sss SPP=256 REPS=2 timeout 600 python scripts/frame_ab.py "" 2>&1 | grep "ms$"
done
} > gpurun_out/r6_walk.txt 2>&1
cat gpurun_out/r6_walk.txt
