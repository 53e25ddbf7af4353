#!/bin/bash
mkdir -p gpurun_out
{
echo "== curve / hair parity with the pretest + exact curve turns (main lib)"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "curve or hair or soup or trace_hooks or bvh or tail or instance" 2>&1 | grep -E "passed|failed|error" | tail -3
echo "== hair frame (128 spp)"
for lib in pbrlab_amd/libpbrhip.so build/p0/libpbrhip.so build/cx2/libpbrhip.so build/s20/libpbrhip.so build/cx2s20/libpbrhip.so; do
  echo "-- $lib"
  PBRHIP_LIB=$(realpath $lib) VARIANT=hair SPP=128 REPS=2 timeout 600 python scripts/frame_ab.py "" 2>&1 | grep "ms$"
done
echo "== hair: per-ray statistics (main lib)"
PBRHIP_PV_STATS=1 VARIANT=hair SPP=8 timeout 600 python scripts/qtree_probe.py 2>&1 | grep -v "amdgpu.ids" | grep "WIDE=1\|closest\|shadow\|pv \|curve leaves\|wide nodes" | head -9
echo "== C5 scene (cornell + hair, 3840x2160, 16 spp): main vs p0"
for lib in pbrlab_amd/libpbrhip.so build/p0/libpbrhip.so; do
  PBRHIP_LIB=$(realpath $lib) VARIANT=c5 SPP=16 timeout 900 python scripts/qtree_probe.py 2>&1 | grep "WIDE=1"
done
} > gpurun_out/r6_hair2.txt 2>&1
cat gpurun_out/r6_hair2.txt
