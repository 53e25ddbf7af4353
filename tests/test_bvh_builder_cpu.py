"""CPU: the host tree builders (pbrlab_amd/csrc/bvh_build.cpp: binned-SAH binary tree, the Q tree collapsed from it) on 399
random and degenerate primitive sets -- scripts/fuzz/bvh_check.cpp compiled for the host only.  Checked there: every primitive
in exactly one leaf of either tree, leaf kinds, every stored box contains its primitives, every QUANTISED child box -- rebuilt
with the traversal's own expression fmaf(q, s, org) -- contains the binary tree's widened box, curve leaves of the Q tree are
chains of neighbouring points, the reported traversal-stack need is the true maximum."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def test_host_builders_on_random_and_degenerate_sets(tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    cs = os.path.join(ROOT, "pbrlab_amd", "csrc")
    exe = str(tmp_path / "bvh_check")
    subprocess.check_call([HIPCC, "--offload-host-only", "-std=c++17", "-O1", "-I" + cs, "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "scripts", "fuzz", "bvh_check.cpp"), os.path.join(cs, "bvh_build.cpp"), "-o", exe, "-lpthread"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "cases ok" in r.stdout, r.stdout + r.stderr
