mkdir -p gpurun_out/r4e
{
PBRHIP_TRACEWP=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4
for lib in pbrlab_amd/libpbrhip.so build/wpB/libpbrhip.so build/wpC/libpbrhip.so build/wpD/libpbrhip.so; do
PBRHIP_LIB=$lib PBRHIP_PV_STATS=1 timeout 300 python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import pbrlab_amd as pa
from pbrlab_amd import scenes
s = pa.scene_from_desc(scenes.cornell_scene("ggx", seed=1))
layer = pa.RenderLayer()
pa.Render(s, 1920, 1080, 8, layer=layer)
for t in ("0", "1", "0", "1"):
    os.environ["PBRHIP_TRACEWP"] = t
    ok, tm = pa.Render(s, 1920, 1080, 8, layer=layer, flags=pa.api.RENDER_TIMING, num_streams=1)
    print("%s TRACEWP=%s 8 spp: frame %.1f ms k_trace %.2f" % (os.environ.get("PBRHIP_LIB"), t, tm["ms_total"], tm["ms_trace_closest"]), flush=True)
ok, st = pa.Render(s, 1920, 1080, 8, layer=layer, flags=pa.api.RENDER_STATS, num_streams=1)
PY
done
timeout 300 scripts/kt.sh base PBRHIP_TRACEWP=0
timeout 300 scripts/kt.sh wp PBRHIP_TRACEWP=1
} > gpurun_out/r4e/wp.log 2>&1
cat gpurun_out/r4e/wp.log
