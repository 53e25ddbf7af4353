#!/bin/bash
mkdir -p gpurun_out
{
for lib in pbrlab_amd/libpbrhip.so build/leaf1/libpbrhip.so build/leaf1one/libpbrhip.so pbrlab_amd/libpbrhip.so build/leaf1/libpbrhip.so build/leaf1one/libpbrhip.so; do
  echo "== $lib"
  PBRHIP_LIB=$(realpath $lib) python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import pbrlab_amd as pa
from pbrlab_amd import scenes
s = pa.scene_from_desc(scenes.hair_scene(seed=1))
layer = pa.RenderLayer()
pa.Render(s, 1920, 1080, 128, layer=layer)
ts = []
for _ in range(3):
    t = time.perf_counter(); pa.Render(s, 1920, 1080, 128, layer=layer); ts.append((time.perf_counter() - t) * 1e3)
ok, tm = pa.Render(s, 1920, 1080, 128, layer=layer, flags=pa.api.RENDER_TIMING, num_streams=1)
ok, st = pa.Render(s, 1920, 1080, 8, layer=layer, flags=pa.api.RENDER_STATS)
cr = max(st["closest_rays"], 1); sr = max(st["shadow_rays"], 1)
print("hair frames", [round(t, 1) for t in ts], "trace", round(tm["ms_trace_closest"], 1), "| closest: nodes/ray %.2f curves %.2f tris %.2f | shadow: nodes %.2f curves %.2f" % (
    st["closest_nodes"] / cr, st["closest_curves"] / cr, st["closest_tris"] / cr, st["shadow_nodes"] / sr, st["shadow_curves"] / sr), "info", {k: v for k, v in s.info().items() if "node" in k or "depth" in k}, flush=True)
PY
done
PBRHIP_LIB=$(realpath build/leaf1/libpbrhip.so) timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider -k "hair or curve or soup" 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tail -3
} > gpurun_out/r6_leaf1.txt 2>&1
cat gpurun_out/r6_leaf1.txt
