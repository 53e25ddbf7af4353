#!/bin/bash
mkdir -p gpurun_out
{
echo "== parity"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "render_matches or random_materials or resumable or degenerate or doomed or textured" 2>&1 | grep -E "passed|failed|error" | tail -3
export REPS=3 SCHED_CONFIGS='[{}]'
for lib in pbrlab_amd/libpbrhip.so build/nopre/libpbrhip.so pbrlab_amd/libpbrhip.so build/nopre/libpbrhip.so; do
  echo "== $lib"
  PBRHIP_LIB=$(realpath $lib) timeout 600 python scripts/sched_ab.py ggx 2>&1 | grep "^{\|^1/8"
  PBRHIP_LIB=$(realpath $lib) VARIANT=ggx SPP=64 REPS=1 timeout 600 python scripts/frame_ab.py "PBRHIP_STREAMS=1" 2>&1 | grep "ms$"
done
} > gpurun_out/r6_shade.txt 2>&1
cat gpurun_out/r6_shade.txt
