"""phase-voting statistics (turns, lanes per turn) of one render with statistics, one ray per lane vs two (PBRHIP_TRACE2)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrlab_amd as pa
from pbrlab_amd import scenes
variant = os.environ.get("VARIANT", "ggx")
spp = int(os.environ.get("SPP", "8"))
desc = scenes.hair_scene(seed=1) if variant == "hair" else scenes.cornell_scene(variant, seed=1)
s = pa.scene_from_desc(desc)
os.environ["PBRHIP_PV_STATS"] = "1"
for t2 in ("0", "1"):
    os.environ["PBRHIP_TRACE2"] = t2
    layer = pa.RenderLayer()
    pa.Render(s, 1920, 1080, spp, layer=layer)
    ok, tm = pa.Render(s, 1920, 1080, spp, layer=layer, flags=pa.api.RENDER_TIMING, num_streams=1)
    print(f"TRACE2={t2} {variant} {spp} spp: frame {tm['ms_total']:.1f} ms, k_trace {tm['ms_trace_closest']:.2f} ms in {tm['n_trace_closest']} launches", flush=True)
    sys.stderr.flush()
    ok, st = pa.Render(s, 1920, 1080, spp, layer=layer, flags=pa.api.RENDER_STATS, num_streams=1)
    sys.stderr.flush()
