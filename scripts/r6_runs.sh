#!/bin/bash
mkdir -p gpurun_out
{
for lib in pbrlab_amd/libpbrhip.so build/runs1/libpbrhip.so pbrlab_amd/libpbrhip.so build/runs1/libpbrhip.so; do
  echo "== $lib"
  PBRHIP_LIB=$(realpath $lib) python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import pbrlab_amd as pa
from pbrlab_amd import scenes
for variant, spp in (("sss", 256), ("hair", 128)):
    desc = scenes.hair_scene(seed=1) if variant == "hair" else scenes.cornell_scene(variant, seed=1)
    s = pa.scene_from_desc(desc)
    layer = pa.RenderLayer()
    pa.Render(s, 1920, 1080, spp, layer=layer)
    ok, tm = pa.Render(s, 1920, 1080, spp, layer=layer, flags=pa.api.RENDER_TIMING, num_streams=1)
    ts = []
    for _ in range(3):
        t = time.perf_counter(); pa.Render(s, 1920, 1080, spp, layer=layer); ts.append((time.perf_counter() - t) * 1e3)
    print(variant, {k[3:]: round(v, 2) for k, v in tm.items() if k.startswith("ms_") and v > 0.5}, "frames", [round(t, 1) for t in ts], flush=True)
PY
done
} > gpurun_out/r6_runs.txt 2>&1
cat gpurun_out/r6_runs.txt
