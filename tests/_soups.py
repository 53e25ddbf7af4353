"""Adversarial primitive sets for the intersection contract: exact duplicates, coplanar overlaps, grid-snapped
shared vertices / edges, zero-area triangles and numerically degenerate slivers (they can pass the triangle test with a
meaningless distance far from the sliver: the case the contract's validation rule exists for).  Shared by the CPU test
(checker tree == checker brute force) and the GPU tests (every GPU tree == checker brute force)."""
import numpy as np

import _oracle as O
from pbrlab_amd import scenes


def triangle_soup(seed, extra_slivers=0):
    """-> (desc, rays).  extra_slivers = 0 is the soup of rounds 1-2 (seed 1, ray 6858 is the documented ray on which a
    brute-force loop and a near-first traversal disagreed before the validation rule); extra_slivers > 0 appends that many
    slivers / needles of several kinds and aims a share of the rays at them."""
    rng = np.random.RandomState(500 + seed)
    n = 400
    v = (rng.rand(n, 3, 3).astype(np.float32) * 2 - 1)
    v[:, 1:] = v[:, :1] + (v[:, 1:] - v[:, :1]) * np.float32(0.3)           # smallish triangles
    v[:60] = np.round(v[:60] * 4) / 4                                        # snapped to a grid: shared vertices / edges
    v[60:100, :, 2] = np.float32(0.25)                                      # coplanar, overlapping
    v[100:120] = v[60:80]                                                    # exact duplicates
    v[120:130, 2] = v[120:130, 1]                                            # degenerate
    v[130:140, 2] = v[130:140, 0] + (v[130:140, 1] - v[130:140, 0]) * np.float32(1.000001)   # slivers
    if extra_slivers:
        r2 = np.random.RandomState(7000 + seed)
        e = (r2.rand(extra_slivers, 3, 3).astype(np.float32) * 2 - 1)
        e[:, 1] = e[:, 0] + (e[:, 1] - e[:, 0]) * np.float32(0.6)
        f = np.float32(1.0) + np.float32(2.0) ** (-r2.randint(12, 24, size=extra_slivers)).astype(np.float32) * r2.choice([-1, 1], size=extra_slivers).astype(np.float32)
        f[::5] = np.float32(0.5)                                             # third corner ON the edge, in its middle
        e[:, 2] = e[:, 0] + (e[:, 1] - e[:, 0]) * f[:, None]
        k = np.arange(extra_slivers) % 3 == 1                                # needles: a hair's breadth off the edge
        e[k, 2] += (r2.rand(int(k.sum()), 3).astype(np.float32) - np.float32(0.5)) * np.float32(1e-6)
        v = np.concatenate([v, e])
        n += extra_slivers
    verts = np.concatenate([v.reshape(-1, 3), np.ones((n * 3, 1), np.float32)], 1)
    faces = np.arange(n * 3, dtype=np.uint32).reshape(n, 3)
    mat = dict(scenes.PRINCIPLED_DEFAULTS, kind="principled", name="m")
    desc = scenes.SceneDesc(verts, np.zeros((0, 4), np.float32), [mat],
                            [scenes.Shape("a", faces[:250], None, np.zeros(250, np.uint32)),
                             scenes.Shape("b", faces[250:], None, np.zeros(n - 250, np.uint32))])
    so = O.oracle_scene_from_desc(desc)
    lo, hi = so.FetchSceneAABB()
    rays = scenes.random_rays((lo, hi), 6000, seed=seed)
    extra = np.zeros(2000, O.RAY_DT)                                         # straight at vertices / along grid lines
    tgt = verts[rng.randint(len(verts), size=2000), :3]
    org = np.array([0.3, -0.2, 3.0], np.float32)
    extra["org"] = org
    extra["dir"] = tgt - org
    extra[:500]["org"] = tgt[:500] + np.array([0, 0, 2], np.float32)
    extra[:500]["dir"] = (0, 0, -1)
    extra["tmin"], extra["tmax"] = 0.0, 1e30
    rays = np.concatenate([rays, extra])
    if extra_slivers:
        # rays through points of the slivers' long edges, from random origins (most pass the sliver's plane test region)
        r3 = np.random.RandomState(9000 + seed)
        m = 3000
        which = r3.randint(400, n, size=m)
        w = r3.rand(m, 1).astype(np.float32)
        tgt = v[which, 0] * (1 - w) + v[which, 1] * w
        aim = np.zeros(m, O.RAY_DT)
        o = (r3.rand(m, 3).astype(np.float32) * 2 - 1) * np.float32(1.5)
        aim["org"], aim["dir"] = o, (tgt - o) * np.float32(0.5 + r3.rand())   # hit distances around 1 / that factor
        aim["tmin"], aim["tmax"] = 0.0, 1e30
        rays = np.concatenate([rays, aim])
    return desc, so, rays
