"""GPU: the library's RCCL exchanges with more than one rank (north_star: "RCCL reduce of the RenderLayer framebuffer over
xGMI").  Runs wherever at least two devices are visible: one FRESH process per device (tests/_rccl_rank.py; nothing is
re-executed from a process that has touched the GPU), world = 2 .. number of devices; the gathered frame and the reduced
frame must equal the one-rank frame bit for bit.  On a one-GPU box it is skipped with the reason printed (RCCL refuses two
ranks on one device); the same pixel lists and both exchanges run at world sizes 2 and 3 over gloo in tests/test_dist_cpu.py."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.mark.gpu
def test_rccl_gather_and_reduce_with_every_world_size():
    import pbrlab_amd as pa
    from pbrlab_amd import scenes
    ndev = pa.device_count()
    if ndev < 2:
        print(f"multi-rank RCCL test skipped: {ndev} device(s) visible, RCCL needs one device per rank")
        pytest.skip(f"{ndev} device(s) visible: RCCL needs one device per rank")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _rccl_rank as R
    pa.set_device(0)
    scene = pa.scene_from_desc(scenes.cornell_hair_scene("sss", n_strands=200, n_segments=5, monkey_subdiv=2, lucy_nu=64, lucy_nv=12))
    one = pa.RenderLayer()
    pa.Render(scene, R.W, R.H, R.SPP, layer=one)
    assert one.rgba[..., :3].max() > 0
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for world in range(2, ndev + 1):
        with tempfile.TemporaryDirectory() as tmp:
            idf, out = os.path.join(tmp, "id"), os.path.join(tmp, "frames.npz")
            procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_rccl_rank.py"), str(r), str(world), idf, out],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
            logs = []
            for p in procs:
                try:
                    logs.append(p.communicate(timeout=600)[0])
                except subprocess.TimeoutExpired:
                    for q in procs:
                        q.kill()
                    raise
            assert all(p.returncode == 0 for p in procs), (world, logs)
            f = np.load(out)
            for exchange in ("gather", "reduce"):
                assert np.array_equal(f[exchange + "_count"], one.count), (world, exchange)
                assert np.array_equal(f[exchange + "_rgba"].view(np.uint32), one.rgba.view(np.uint32)), (world, exchange)
        print(f"world {world}: gathered and reduced frames equal the one-rank frame")
