#!/bin/bash
mkdir -p gpurun_out
{
echo "== parity (main lib: eight queue heads, batches <= 128)"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "render_matches or resumable or patch_order or random_materials or trace_hooks or concurrent or group_schedules or sharding" 2>&1 | grep -E "passed|failed|error" | tail -3
export REPS=3 SCHED_CONFIGS='[{}]'
for lib in build/h1/libpbrhip.so pbrlab_amd/libpbrhip.so build/hb256/libpbrhip.so build/hb64/libpbrhip.so build/hb128g4/libpbrhip.so build/h1/libpbrhip.so pbrlab_amd/libpbrhip.so; do
  echo "== $lib"
  PBRHIP_LIB=$(realpath $lib) timeout 600 python scripts/sched_ab.py ggx 2>&1 | grep "^{\|^1/8\|^!!"
done
echo "== wave log, eighth of C2 (main lib)"
timeout 300 python scripts/wave_log.py 8 2>&1 | grep launch
echo "== hair / sss frames: one head vs eight"
for lib in build/h1/libpbrhip.so pbrlab_amd/libpbrhip.so; do
  PBRHIP_LIB=$(realpath $lib) VARIANT=hair SPP=128 REPS=2 timeout 600 python scripts/frame_ab.py "" 2>&1 | grep "ms$"
  PBRHIP_LIB=$(realpath $lib) VARIANT=sss SPP=256 REPS=2 timeout 600 python scripts/frame_ab.py "" 2>&1 | grep "ms$"
done
} > gpurun_out/r6_heads.txt 2>&1
cat gpurun_out/r6_heads.txt
