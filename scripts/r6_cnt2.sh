#!/bin/bash
mkdir -p gpurun_out
{
for lib in pbrlab_amd/libpbrhip.so build/ci4/libpbrhip.so build/ci16/libpbrhip.so build/co16/libpbrhip.so build/co8/libpbrhip.so pbrlab_amd/libpbrhip.so; do
  echo "== $lib"
  PBRHIP_LIB=$(realpath $lib) python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import pbrlab_amd as pa
from pbrlab_amd import scenes
for variant, spp in (("ggx", 64), ("sss", 256)):
    s = pa.scene_from_desc(scenes.cornell_scene(variant, seed=1))
    layer = pa.RenderLayer()
    pa.Render(s, 1920, 1080, spp, layer=layer, num_streams=1)
    best = None
    for _ in range(2):
        ok, tm = pa.Render(s, 1920, 1080, spp, layer=layer, flags=pa.api.RENDER_TIMING, num_streams=1)
        if best is None or tm["ms_surface"] + tm["ms_compact"] < best["ms_surface"] + best["ms_compact"]:
            best = tm
    print(variant, {k[3:]: round(v, 2) for k, v in best.items() if k in ("ms_surface", "ms_compact", "ms_shade_principled", "ms_trace_closest")})
PY
done
} > gpurun_out/r6_cnt2.txt 2>&1
cat gpurun_out/r6_cnt2.txt
