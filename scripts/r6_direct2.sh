#!/bin/bash
mkdir -p gpurun_out
{
for cfg in "build/head/libpbrhip.so 0" "pbrlab_amd/libpbrhip.so 0" "pbrlab_amd/libpbrhip.so 1" "build/head/libpbrhip.so 0" "pbrlab_amd/libpbrhip.so 0" "pbrlab_amd/libpbrhip.so 1"; do
  set -- $cfg
  echo "== $1 PBRHIP_DIRECT=$2"
  PBRHIP_LIB=$(realpath $1) PBRHIP_DIRECT=$2 python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import pbrlab_amd as pa
from pbrlab_amd import scenes
s = pa.scene_from_desc(scenes.cornell_scene("ggx", seed=1))
layer = pa.RenderLayer()
pa.Render(s, 1920, 1080, 64, layer=layer, num_streams=1)
best = None
for _ in range(3):
    ok, tm = pa.Render(s, 1920, 1080, 64, layer=layer, flags=pa.api.RENDER_TIMING, num_streams=1)
    if best is None or tm["ms_total"] < best["ms_total"]: best = tm
print({k[3:]: round(v, 2) for k, v in best.items() if k.startswith("ms_") and v})
ts = []
for _ in range(4):
    t = time.perf_counter(); pa.Render(s, 1920, 1080, 64, layer=layer); ts.append((time.perf_counter() - t) * 1e3)
print("frame (default groups, with the layer copy) ms:", [round(t, 2) for t in ts])
PY
done
PBRHIP_DIRECT=1 timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -4
} > gpurun_out/r6_direct2.txt 2>&1
cat gpurun_out/r6_direct2.txt
