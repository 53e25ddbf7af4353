#!/usr/bin/env python3
"""Measures -- instead of projecting -- how the difference between the GPU frame and the reference's own arithmetic
(oracle[libm]: the host libm's cosf / sinf / expf / logf) falls with the sample count on BASELINE configs[4] (C5: S-cornell
with subsurface + S-hair, 3840 x 2160).  A last-ulp difference between libm and the f64r functions flips a discrete decision
of a sample now and then; the flipped samples are independent, so the relative L2 of the frame should
fall as 1 / sqrt(spp).  Renders the frame at 1, 4 and 16 spp on the GPU and with oracle[libm] on the host cores and records
relative L2, the differing / flipped pixels and the fitted exponent; also checks GPU == oracle[f64r] bit for bit at 1 spp.

usage (GPU box):  python scripts/c5_spp_law.py [--spps 1,4,16] [--scale 1.0]   -> gpurun_out/profiles/r3_c5_spp_law.json"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O  # noqa: E402
import pbrlab_amd as pa  # noqa: E402
from pbrlab_amd import scenes  # noqa: E402


def rel_l2(a, b):
    return float(np.linalg.norm((a[..., :3] - b[..., :3]).astype(np.float64)) / np.linalg.norm(b[..., :3].astype(np.float64)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spps", default="1,4,16")
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of 3840 x 2160 per axis (1.0 = the configuration)")
    args = ap.parse_args()
    W, H = int(3840 * args.scale), int(2160 * args.scale)
    threads = os.cpu_count() or 8
    desc = scenes.cornell_hair_scene("sss", seed=1)
    sg, so = pa.scene_from_desc(desc), O.oracle_scene_from_desc(desc)
    rows = []
    for spp in [int(s) for s in args.spps.split(",")]:
        lay = pa.RenderLayer()
        t0 = time.time()
        pa.Render(sg, W, H, spp, layer=lay)
        t_gpu = time.time() - t0
        t0 = time.time()
        libm, cnt, _ = so.render(W, H, spp, threads=threads, math_mode=O.MATH_LIBM)
        t_cpu = time.time() - t0
        assert np.array_equal(lay.count, cnt)
        mean_g, mean_l = lay.rgba[..., :3] / spp, libm[..., :3] / spp
        d = np.abs(mean_g - mean_l).max(axis=2)
        flipped = d > 1e-3 * np.maximum(mean_l.max(axis=2), 0.05) / spp      # one flipped sample moves the mean by O(1 / spp)
        row = {"spp": spp, "rel_l2": rel_l2(lay.rgba, libm), "pixels_differing": int((d > 0).sum()), "pixels_with_a_flipped_sample": int(flipped.sum()),
               "rel_l2_without_them": float(np.linalg.norm((mean_g - mean_l)[~flipped].astype(np.float64)) / np.linalg.norm(mean_l[~flipped].astype(np.float64))),
               "gpu_s": round(t_gpu, 2), "oracle_libm_s": round(t_cpu, 1)}
        if spp == 1:
            f64r, _, _ = so.render(W, H, 1, threads=threads, math_mode=O.MATH_F64R)
            row["pixels_differing_from_oracle_f64r"] = int((lay.rgba != f64r).any(axis=2).sum())
        rows.append(row)
        print(row, flush=True)
    xs, ys = np.log([r["spp"] for r in rows]), np.log([r["rel_l2"] for r in rows])
    slope = float(np.polyfit(xs, ys, 1)[0]) if len(rows) > 1 else None
    out = {"config": "c5", "width": W, "height": H, "cpu_threads": threads, "rows": rows, "fitted_exponent": slope,
           "expected_exponent": -0.5, "projected_rel_l2_at_1024_spp_from_the_fit": float(np.exp(np.polyval(np.polyfit(xs, ys, 1), np.log(1024)))) if len(rows) > 1 else None,
           "bar": 1e-4}
    os.makedirs(os.path.join(ROOT, "gpurun_out", "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "profiles", "r3_c5_spp_law.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "rows"}))


if __name__ == "__main__":
    main()
