#!/bin/bash
mkdir -p gpurun_out
{
python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import pbrlab_amd as pa
from pbrlab_amd import scenes
for variant, spp in (("sss", 256), ("hair", 128)):
    desc = scenes.hair_scene(seed=1) if variant == "hair" else scenes.cornell_scene(variant, seed=1)
    s = pa.scene_from_desc(desc)
    layer = pa.RenderLayer()
    pa.Render(s, 1920, 1080, spp, layer=layer)
    for rep in range(2):
        for tp in ("65536", "131072", "262144", "524288", "1048576", "2097152"):
            os.environ["PBRHIP_TAIL_PATHS"] = tp
            ts = []
            for _ in range(3):
                t = time.perf_counter(); pa.Render(s, 1920, 1080, spp, layer=layer); ts.append((time.perf_counter() - t) * 1e3)
            print(variant, "tail_paths", tp, [round(t, 1) for t in ts], flush=True)
    os.environ.pop("PBRHIP_TAIL_PATHS")
PY
} > gpurun_out/r6_tailsweep.txt 2>&1
cat gpurun_out/r6_tailsweep.txt
