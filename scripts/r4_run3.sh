mkdir -p gpurun_out/r4c
{
PBRHIP_TRACEQ=1 PBRHIP_PV_STATS=1 SPP=8 timeout 300 python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import pbrlab_amd as pa
from pbrlab_amd import scenes
s = pa.scene_from_desc(scenes.cornell_scene("ggx", seed=1))
layer = pa.RenderLayer()
pa.Render(s, 1920, 1080, 8, layer=layer)
ok, tm = pa.Render(s, 1920, 1080, 8, layer=layer, flags=pa.api.RENDER_TIMING, num_streams=1)
print("pool 8 spp: frame %.1f ms k_trace %.2f" % (tm["ms_total"], tm["ms_trace_closest"]), flush=True)
ok, st = pa.Render(s, 1920, 1080, 8, layer=layer, flags=pa.api.RENDER_STATS, num_streams=1)
print({k: st[k] for k in ("closest_rays", "shadow_rays")})
PY
SPP=8 PBRHIP_TRACEQ=1 PASS_TIMEOUT=200 timeout 900 python scripts/pmc_adhoc.py c2 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" 2>&1 | grep -v "^$" | grep "pass\|k_trace"
} > gpurun_out/r4c/pool.log 2>&1
cat gpurun_out/r4c/pool.log
