mkdir -p gpurun_out/r4o
{
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -5
for i in 1 2; do
timeout 300 scripts/kt.sh base PBRHIP_LIB=build/base/libpbrhip.so
timeout 300 scripts/kt.sh new
done
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import pbrlab_amd as pa
from pbrlab_amd import scenes
s = pa.scene_from_desc(scenes.cornell_scene("ggx", seed=1))
layer = pa.RenderLayer()
pa.Render(s, 1920, 1080, 64, layer=layer)
for rep in range(2):
    best = min(pa.Render(s, 1920, 1080, 64, layer=layer)[1]["ms_total"] for _ in range(5))
    print("frame (two groups) best of 5: %.2f ms" % best, flush=True)
PY
} 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" > gpurun_out/r4o/gen.log
cat gpurun_out/r4o/gen.log
