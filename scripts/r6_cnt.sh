#!/bin/bash
mkdir -p gpurun_out
{
echo "== parity (counters spread over cache lines)"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "render_matches or resumable or random_materials or trace_hooks or concurrent or group_schedules or hair or sss" 2>&1 | grep -E "passed|failed|error" | tail -3
for round in 1 2; do
for lib in build/packed/libpbrhip.so pbrlab_amd/libpbrhip.so; do
  echo "== $lib"
  PBRHIP_LIB=$(realpath $lib) VARIANT=ggx SPP=64 REPS=1 timeout 600 python scripts/frame_ab.py "" "PBRHIP_STREAMS=1" 2>&1 | grep "ms$"
  PBRHIP_LIB=$(realpath $lib) VARIANT=sss SPP=256 REPS=1 timeout 600 python scripts/frame_ab.py "" 2>&1 | grep "ms$"
  PBRHIP_LIB=$(realpath $lib) VARIANT=hair SPP=128 REPS=1 timeout 600 python scripts/frame_ab.py "" 2>&1 | grep "ms$"
  PBRHIP_LIB=$(realpath $lib) python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import pbrlab_amd as pa
from pbrlab_amd import scenes
for variant, spp in (("ggx", 64), ("sss", 256)):
    s = pa.scene_from_desc(scenes.cornell_scene(variant, seed=1))
    layer = pa.RenderLayer()
    pa.Render(s, 1920, 1080, spp, layer=layer, num_streams=1)
    ok, tm = pa.Render(s, 1920, 1080, spp, layer=layer, flags=pa.api.RENDER_TIMING, num_streams=1)
    print(variant, {k[3:]: round(v, 2) for k, v in tm.items() if k.startswith("ms_") and v})
PY
done
done
} > gpurun_out/r6_cnt.txt 2>&1
cat gpurun_out/r6_cnt.txt
