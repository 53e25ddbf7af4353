#!/usr/bin/env python3
"""bench.py -- Msamples/s of the MI355X path-tracing hot path (BASELINE.json metric).

One "step" = one pbrlab::Render() of the workload: every (pixel, pass) sample of the frame goes
through the whole hot path (camera ray -> GetRadiance bounce loop -> RenderLayer).  The scene (BVH,
materials, light tables) is already resident in HBM when the timed region starts; the step ends with
the framebuffer on the host of rank 0 (RenderLayer is host memory in pbrlab's API).

N GPUs: one process per GPU (launched by torch.distributed.run; torch.distributed is used for the rendezvous, the
barrier and the max-over-ranks of the time).  16x16 pixel blocks are dealt to the ranks (block i -> rank i % N), every
rank renders its blocks of the SAME 1920x1080 x 64 spp frame into a zeroed full-size device framebuffer ("scaling":
"strong": BASELINE's metric is this frame at 1/2/4/8 GPUs), and the library's own RCCL communicator (pbrhip_comm_*,
include/pbrhip.h) lands the frame on rank 0: by default a gather of the ranks' shards (every shard over its own xGMI link;
--exchange reduce = ncclReduce of the full layers, --exchange torch = torch.distributed.reduce).  The weak-scaling figure
(spp = 64 x N, per-GPU work fixed) is measured in the same run and reported as the secondary object "weak".

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[1..4] on the procedural stand-in scenes (assets are missing: BASELINE.md §3)
    "c2": dict(desc="cornellbox_suzanne_lucy stand-in (S-cornell, reference .mtl, Lucy subsurface=0): "
                    "Lambert + GGX, 1920x1080, 64 spp", scene="cornell", variant="ggx", width=1920, height=1080, spp=64),
    "c3": dict(desc="S-cornell with Lucy subsurface=1 (random-walk SSS), 1920x1080, 256 spp", scene="cornell",
               variant="sss", width=1920, height=1080, spp=256),
    "c4": dict(desc="S-hair (50k curly strands) + head, PrincipledHair + curve intersection, 1920x1080, 128 spp",
               scene="hair", variant="", width=1920, height=1080, spp=128),
    "c1": dict(desc="S-cornell Lambert-only, 256x256, 4 spp (plumbing config)", scene="cornell", variant="lambert",
               width=256, height=256, spp=4),
    "c5": dict(desc="S-cornell (SSS) + S-hair, 3840x2160, 1024 spp (BASELINE configs[4]; 8.5 G samples: meant for 8 GPUs)",
               scene="cornell_hair", variant="sss", width=3840, height=2160, spp=1024),
}

# algorithmic bytes of the traversal kernel k_trace: 64 B per BVH node visited, 48 B per
# triangle tested, 64 B per curve tested; per closest-hit ray 32 B ray + 16 B hit record + 4 B queue entry; per
# shadow ray 32 B ray (origin shared with the continuation ray) + 4 B queue entry + 16 B pending contribution
NODE_B, TRI_B, CURVE_B, RAY_B, SHADOW_RAY_B = 64, 48, 64, 52, 52
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def pmc_traffic(workload, spp, world):
    """HBM bytes per k_trace launch from the committed PMC passes (profiles/README.md: FETCH_SIZE and WRITE_SIZE in
    separate --pmc runs of the same render, (2*FETCH_SIZE + WRITE_SIZE) * 1024 with the gfx950 FETCH_SIZE x2
    correction of MI355X_MICROARCH.md, calibrated on k_accumulate's known 2.12 GB stream).  Counters cannot be read
    live, so this is only reported for the exact configuration they were collected on; otherwise null."""
    path = os.path.join(ROOT, "profiles", "r1_c2_hbm_traffic_pmc.json")
    if workload != "c2" or spp != WORKLOADS["c2"]["spp"] or world != 1 or not os.path.exists(path):
        return None
    rec = json.load(open(path)).get("pb::k_trace<false, false>")
    return rec["hbm_bytes_per_dispatch_fetch_x2"] if rec else None


def make_desc(w):
    from pbrlab_amd import scenes
    if w["scene"] == "cornell":
        return scenes.cornell_scene(w["variant"], seed=1)
    if w["scene"] == "cornell_hair":
        return scenes.cornell_hair_scene(w["variant"], seed=1)
    return scenes.hair_scene(seed=1)


def cpu_baseline(desc, w, seconds_budget=20.0):
    """Oracle (reference algorithm restated in C, oracle/) on the host cores, bounded sample of the same
    workload: the full frame at as many spp as fit ~seconds_budget (cost per sample is spp-independent)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    so = O.oracle_scene_from_desc(desc)
    cores = os.cpu_count() or 1
    W, H = w["width"], w["height"]
    t0 = time.time()                               # calibrate on 1/16 of the tiles
    _, _, st = so.render(W, H, 1, tile_rank=0, tile_world=16, threads=cores)
    rate = st["samples"] / max(time.time() - t0, 1e-3)
    spp, world = 1, 1
    if rate * seconds_budget >= W * H:
        spp = int(max(1, min(w["spp"], rate * seconds_budget // (W * H))))
    else:
        world = int(min(64, max(1, round(W * H / (rate * seconds_budget)))))
    t0 = time.time()
    _, _, st = so.render(W, H, spp, tile_rank=0, tile_world=world, threads=cores)
    dt = time.time() - t0
    return {"value": st["samples"] / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"oracle (C restatement of the reference integrator, own binned-SAH BVH2; Embree is not "
                      f"available), {W}x{H} x {spp} spp, tiles i%{world}==0: {st['samples']} samples in {dt:.1f} s "
                      f"on {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override spp (invalidates the headline config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--max-paths", type=int, default=0)
    ap.add_argument("--streams", type=int, default=0, help="concurrent path groups (0 = the library's default)")
    ap.add_argument("--shard-block", type=int, default=16,
                    help="edge of the pixel blocks dealt to the ranks at N > 1 (the reference's tile is 64; smaller balances better)")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="N > 1: strong = the fixed 64-spp frame split over N GPUs (default: BASELINE's metric); "
                         "weak = spp x N (per-GPU work fixed).  The other one is reported as a secondary object")
    ap.add_argument("--exchange", default="gather", choices=["gather", "reduce", "torch"],
                    help="N > 1: how the RenderLayer reaches rank 0: gather / reduce = RCCL inside libpbrhip "
                         "(pbrhip_comm_gather_layer / pbrhip_comm_reduce_layer), torch = torch.distributed.reduce")
    args = ap.parse_args()

    import numpy as np
    import torch
    import pbrlab_amd as pa
    from pbrlab_amd import api
    from pbrlab_amd.dist import reduce_layer

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    pa.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    w = dict(WORKLOADS[args.workload])
    if args.spp:
        w["spp"] = args.spp
    W, H = w["width"], w["height"]
    weak = args.scaling == "weak"
    spp = w["spp"] * (world if weak else 1)   # strong (default): the frame is fixed; weak: W*H/N pixels x 64*N spp per GPU
    desc = make_desc(w)
    scene = pa.scene_from_desc(desc)          # upload + BVH: outside the timed region
    info = scene.info()

    rgba = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    count = torch.zeros((H, W), dtype=torch.int32, device=dev)
    h_rgba = torch.zeros((H, W, 4), dtype=torch.float32).pin_memory() if rank == 0 else None
    h_count = torch.zeros((H, W), dtype=torch.int32).pin_memory() if rank == 0 else None
    torch.cuda.synchronize()
    ptrs = (rgba.data_ptr(), count.data_ptr())

    shard_block = args.shard_block if world > 1 else 0   # one rank: the library's default order (64 x 64 tiles)

    # the exchange step: the library's RCCL communicator (one rank makes the id, torch.distributed hands it round)
    comm, exchange = None, args.exchange
    if dist is not None and exchange != "torch":
        try:                                 # every rank checks that the library finds its RCCL before any collective call
            my_id, usable = pa.Comm.unique_id(), 1
        except Exception as e:
            print(f"bench: library communicator unavailable on rank {rank} ({e})", file=sys.stderr)
            my_id, usable = None, 0
        flag = torch.tensor([usable], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()):
            uid = [my_id if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            comm = pa.Comm(uid[0], rank, world)      # ncclCommInitRank: collective
        else:
            exchange = "torch"                       # reported in config.exchange

    def step(spp, flags=0):
        _, st = api.Render(scene, W, H, spp, tile_rank=rank, tile_world=world, device_out=ptrs, flags=flags,
                           max_paths_in_flight=args.max_paths, num_streams=args.streams, shard_block=shard_block)
        if dist is not None:                   # the only exchange step: the framebuffer, over xGMI
            if exchange == "gather":
                comm.gather_layer(scene, W, H, ptrs[0], ptrs[1], shard_block=shard_block, root=0)
            elif exchange == "reduce":
                comm.reduce_layer(ptrs[0], ptrs[1], W * H, root=0)
            else:
                reduce_layer(rgba, count, dst=0)
        if rank == 0:                          # RenderLayer lives on the host
            h_rgba.copy_(rgba, non_blocking=True)
            h_count.copy_(count, non_blocking=True)
        torch.cuda.synchronize()
        return st

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(spp, steps, warmup, flags):
        for _ in range(warmup):
            step(spp)
        barrier()
        t0 = time.perf_counter()
        agg = {}
        for _ in range(steps):
            st = step(spp, flags=flags)
            for k, v in st.items():
                agg[k] = agg.get(k, 0) + v
        barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        if rank == 0:   # sanity: every pixel got spp samples
            assert int(h_count.min()) == spp and int(h_count.max()) == spp, "framebuffer incomplete"
            assert bool(torch.isfinite(h_rgba).all())
        return elapsed, agg

    elapsed, agg = timed(spp, args.steps, args.warmup, 0 if args.no_roofline else api.RENDER_TIMING)
    other = None
    if world > 1:   # the other scaling mode, same run, as a secondary figure
        o_spp = w["spp"] * (1 if weak else world)
        o_elapsed, _ = timed(o_spp, args.steps, 1, 0)
        other = {"scaling": "strong" if weak else "weak", "spp": o_spp, "ms_per_step": o_elapsed / args.steps * 1e3,
                 "value": W * H * o_spp * args.steps / o_elapsed / 1e6, "unit": "Msamples/s"}

    roofline = None
    if not args.no_roofline:
        # untimed pass with traversal counters (deterministic: identical work to the timed steps)
        _, sst = api.Render(scene, W, H, spp, tile_rank=rank, tile_world=world, device_out=ptrs,
                            flags=api.RENDER_STATS, max_paths_in_flight=args.max_paths, num_streams=args.streams, shard_block=shard_block)
        torch.cuda.synchronize()
        # untimed: the same frame with ONE path group, i.e. every k_trace launch has the GPU to itself
        _, solo = api.Render(scene, W, H, spp, tile_rank=rank, tile_world=world, device_out=ptrs,
                             flags=api.RENDER_TIMING, max_paths_in_flight=args.max_paths, num_streams=1, shard_block=shard_block)
        torch.cuda.synchronize()
        bytes_step = (NODE_B * (sst["closest_nodes"] + sst["shadow_nodes"]) + TRI_B * (sst["closest_tris"] + sst["shadow_tris"]) +
                      CURVE_B * (sst["closest_curves"] + sst["shadow_curves"]) + RAY_B * sst["closest_rays"] +
                      SHADOW_RAY_B * sst["shadow_rays"])
        launches = agg["n_trace_closest"] / args.steps
        ms_step = agg["ms_trace_closest"] / args.steps
        if ms_step > 0:
            achieved = bytes_step / (ms_step * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": "k_trace (closest-hit rays of bounce k + shadow rays of bounce k-1)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "note": "achieved = ALGORITHMIC bytes (64 B per node visit, 48 B per triangle test, 52 B per ray) / kernel time as "
                                "SURVEY 8d defines it; the 128 MB scene is served from L2 / Infinity Cache, so this can exceed the HBM "
                                "peak -- the HBM-side bytes per launch measured with PMC counters are `traffic`.  By default the "
                                "frame runs as two path groups on two HIP streams: their k_trace launches overlap each other and the "
                                "other group's shading, so a launch's duration (HIP events on its own stream, timed region) includes "
                                "time in which it shares the GPU; `solo` is the same frame run as one group (untimed extra render), "
                                "every launch alone on the GPU",
                        "solo": {"launches_per_step": solo["n_trace_closest"], "avg_launch_ms": solo["ms_trace_closest"] / max(solo["n_trace_closest"], 1),
                                 "achieved": bytes_step / (solo["ms_trace_closest"] * 1e-3) / 1e9,
                                 "frac": bytes_step / (solo["ms_trace_closest"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms_frame": solo["ms_total"]},
                        "traffic": None if (args.max_paths or args.streams) else pmc_traffic(args.workload, spp, world),
                        "algorithmic_bytes_per_launch": bytes_step / max(launches, 1),
                        "avg_launch_ms": ms_step / max(launches, 1), "launches_per_step": launches,
                        "rays_per_step": sst["closest_rays"] + sst["shadow_rays"],
                        "kernel_ms_per_step": {{"trace_closest": "trace", "surface": "classify"}.get(k[3:], k[3:]): agg[k] / args.steps
                                               for k in agg if k.startswith("ms_") and k not in ("ms_total",)}}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(desc, w)

    if rank == 0:
        samples = W * H * spp * args.steps
        out = {
            "metric": "Msamples/s (paths x spp / s), 1920x1080 Cornell-box-Suzanne render", "value": samples / elapsed / 1e6,
            "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": w["desc"] + (f" x {world} (spp scaled with the GPU count)" if (world > 1 and weak) else ""), "width": W, "height": H, "spp": spp,
                       "exchange": None if world == 1 else {"gather": "pbrhip_comm_gather_layer (RCCL send/recv of the ranks' shards)",
                                                            "reduce": "pbrhip_comm_reduce_layer (ncclReduce f32 + u32)",
                                                            "torch": "torch.distributed.reduce"}[exchange],
                       "triangles": desc.num_triangles(), "curve_segments": desc.num_segments(),
                       "bvh_nodes": info["num_nodes"], "bvh_depth": info["depth"], "scene_bytes": info["device_bytes"],
                       "parallelism": f"{shard_block}x{shard_block} pixel blocks, block index % {world}" if world > 1 else "1gpu",
                       "rng": "PCG32((pass<<32)+pixel, 1234567890)"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if other is not None:
            out["weak" if not weak else "strong"] = other
        print(json.dumps(out))
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
