"""how the oracle scales with threads on this host (cpu_baseline's honesty check): Msamples/s for 1 .. N threads, the cgroup's CPU quota"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O
from pbrlab_amd import scenes
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except OSError: pass
so = O.oracle_scene_from_desc(scenes.cornell_scene("ggx", seed=1))
W, H = 1920, 1080
for th in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    if th > (os.cpu_count() or 1): break
    world = max(1, 48 // th)
    t0 = time.time(); c0 = time.process_time()
    _, _, st = so.render(W, H, 1, tile_rank=0, tile_world=world, threads=th, job_mode=O.JOBS_TILE_PASS)
    dt = time.time() - t0; cpu = time.process_time() - c0
    print(f"threads {th:3d}: {st['samples'] / dt / 1e6:7.3f} Msamples/s  ({st['samples']} samples, {dt:.2f} s wall, {cpu:.1f} s CPU = {cpu / dt:.1f} cores busy)", flush=True)
