#!/usr/bin/env python3
"""BASELINE configs[0] (C1): S-cornell, Lambert-only closures, 256 x 256, 4 spp, rendered by the ORACLE (the C
restatement of the reference path, oracle/) in both math modes -> tests/golden/c1_oracle.npz.  Like oracle_images.npz this
pins the oracle against drift; it is not a reference output."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O  # noqa: E402
from pbrlab_amd import scenes  # noqa: E402

if __name__ == "__main__":
    desc = scenes.cornell_scene("lambert", seed=1)
    so = O.oracle_scene_from_desc(desc)
    out = {}
    for name, mode in (("f64r", O.MATH_F64R), ("libm", O.MATH_LIBM)):
        rgba, cnt, st = so.render(256, 256, 4, threads=os.cpu_count(), math_mode=mode)
        assert (cnt == 4).all()
        out[f"{name}_rgb"] = rgba[..., :3].copy()
        out[f"{name}_rays"] = np.array([st["closest_rays"], st["shadow_rays"]], np.uint64)
    # the libm frame is stored as its difference from the f64r frame (a few hundred pixels differ, mostly in the last bit)
    a, b = out["f64r_rgb"].reshape(-1, 3), out.pop("libm_rgb").reshape(-1, 3)
    idx = np.nonzero((a != b).any(axis=1))[0]
    out["libm_idx"], out["libm_val"] = idx.astype(np.uint32), b[idx]
    np.savez_compressed(os.path.join(HERE, "c1_oracle.npz"), **out)
    print({k: (v.shape, float(v.sum())) for k, v in out.items()})
