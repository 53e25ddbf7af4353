#!/bin/bash
mkdir -p gpurun_out
{
echo "== curve / hair parity, records with one piece per turn (main lib)"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "curve or hair or soup or trace_hooks or bvh or tail or instance or resumable" 2>&1 | grep -E "passed|failed|error" | tail -3
echo "== hair frame (128 spp): one piece per turn (6 blocks) / both in one turn (5 blocks)"
for lib in pbrlab_amd/libpbrhip.so build/two/libpbrhip.so pbrlab_amd/libpbrhip.so build/two/libpbrhip.so; do
  echo "-- $lib"
  PBRHIP_LIB=$(realpath $lib) VARIANT=hair SPP=128 REPS=2 timeout 600 python scripts/frame_ab.py "" 2>&1 | grep "ms$"
done
echo "== C5 split, 64 spp"
for lib in pbrlab_amd/libpbrhip.so build/two/libpbrhip.so; do
  echo "-- $lib"
  PBRHIP_LIB=$(realpath $lib) SPP=64 timeout 900 python scripts/c5_split.py 2>&1 | tail -2 | cut -c1-330
done
} > gpurun_out/r6_hair4.txt 2>&1
cat gpurun_out/r6_hair4.txt
