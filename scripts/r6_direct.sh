#!/bin/bash
mkdir -p gpurun_out
{
export SCHED_CONFIGS='[{}, {"PBRHIP_DIRECT":"1"}, {}, {"PBRHIP_DIRECT":"1"}]'
REPS=4 python scripts/sched_ab.py ggx 2>&1 | grep -v "^\[sched\]\|^  " | cut -c1-400
for d in 0 1; do
echo "== PBRHIP_DIRECT=$d per-kernel (one group)"
PBRHIP_DIRECT=$d python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import pbrlab_amd as pa
from pbrlab_amd import scenes
s = pa.scene_from_desc(scenes.cornell_scene("ggx", seed=1))
layer = pa.RenderLayer()
pa.Render(s, 1920, 1080, 64, layer=layer, num_streams=1)
for _ in range(2):
    ok, tm = pa.Render(s, 1920, 1080, 64, layer=layer, flags=pa.api.RENDER_TIMING, num_streams=1)
    print({k[3:]: round(v, 2) for k, v in tm.items() if k.startswith("ms_")})
PY
done
PBRHIP_DIRECT=1 timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
} > gpurun_out/r6_direct.txt 2>&1
cat gpurun_out/r6_direct.txt
