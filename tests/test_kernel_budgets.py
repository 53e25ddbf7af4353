"""Register / scratch budgets of the hot kernels, read from the gfx950 code object inside the built libpbrhip.so (no GPU needed).

The traversal kernels live on exact budgets: `k_trace` must fit its waves-per-SIMD target without a byte of scratch (with any
scratch in the loop it is ~30 % slower), and `k_sss_walk` runs without scratch at three waves per SIMD -- at four, 28 bytes more of
it in the traversal loop, caused by moving one statement in dtrace_pv.h, had cost C3 20 % (profiles/README.md).  These limits are the
measured-good values; a change that exceeds one has to be re-measured on the GPU before the limit moves."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"

# kernel (demangled prefix) -> (max VGPRs, max scratch bytes)
BUDGETS = {
    "k_trace<false, false, false, false>": (72, 0),    # binary tree, triangles: 7 waves per SIMD
    "k_trace<false, true, false, false>": (80, 0),     # binary tree, curves: 6 waves
    "k_trace<false, false, true, false>": (80, 0),     # Q tree, triangles: 6 waves
    "k_trace<false, true, true, false>": (80, 0),      # Q tree, curves: 6 waves
    # a group's first launch (camera rays computed in the refill path): one block per CU fewer, nothing spilled into the loop
    "k_trace<false, false, true, true>": (96, 16),     # (16 B: a 12-byte stack object of the packed leaf test, not a spill)
    "k_trace<false, true, true, true>": (96, 0),
    "k_sss_walk<false, false, true>": (168, 24),  # 3 waves per SIMD; 24 B since the packed two-triangle leaf test (round 4): measured 31.1 -> 29.6 ms per 64 spp of C3 WITH them
    "k_sss_walk<false, false, false>": (168, 0),
    "k_sss_walk<false, true, false>": (168, 0),
    "k_sss_walk<false, true, true>": (168, 0),
    "k_shade_principled<0>": (168, 0),           # no medium, no texture (C2, C4)
    # the kernels below keep scratch at three waves per SIMD, measured against two waves without it (profiles/README.md: the
    # general shading kernel alone is 9 % faster at two, the C3 frame 2 % slower; k_tail likewise in round 2): pinned as they are
    "k_shade_principled<1>": (168, 28),          # media, no texture (C3, C5): the medium's coefficients come from the material record
    "k_shade_principled<2>": (168, 108),         # textured materials: ParamToBsdf and the medium per hit
    # (triangle-only scenes: + 16-32 B with the octets of round 4 -- eight lanes per path once a wave has at most eight left,
    # dtrace_quad.h; measured WITH them: k_tail of an eighth of C2 1.59 -> 1.30 ms, the frame 51.0 -> 50.5 ms on the same box)
    "k_tail<0, false, false, true>": (168, 84),    # no medium, no texture (C2); 20 -> 52 B with the packed two-triangle leaf test (C2 k_tail 3.7-4.2 -> 3.5-3.8 ms)
    "k_tail<0, false, true, true>": (168, 28),     # ... with curves (C4); 20 -> 28 B with the camera sample in the shading head (round 4)
    "k_tail<1, false, false, true>": (168, 216),   # media (C3)
    "k_tail<1, false, true, true>": (168, 184),    # media + curves (C5); 168 -> 184 B with the interleaved path-state records (round 4: C3 frame 327 -> 311 ms with them)
    "k_tail<2, false, false, true>": (168, 244),   # textured materials
    "k_trace_quad<false>": (128, 0),             # one ray per quad of lanes (small launches; off by default)
    # the O tree's kernels (round 5, off by default): five blocks per CU (96 VGPRs), nothing spilled
    "k_trace8<false, false, false>": (96, 0),
    "k_trace8<false, true, false>": (96, 0),
    "k_sss_walk8<false, false>": (168, 0),
    "k_shade_hair": (136, 0),
    "k_sss_step": (208, 0),  # 156 -> 201 VGPRs with the packed light pretest (round 4): 3.55 ms per 64 spp of C3 before and after
    "k_classify": (64, 0),
    "k_compact": (96, 0),
}


def kernel_table():
    lib = os.path.join(ROOT, "pbrlab_amd", "libpbrhip.so")
    if not (os.path.exists(lib) and os.path.exists(os.path.join(LLVM, "clang-offload-bundler")) and shutil.which("c++filt")):
        pytest.skip("built library or LLVM tools not available")
    tmp = tempfile.mkdtemp()
    try:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", lib], check=True, capture_output=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    table, cur = {}, {}
    for line in notes.splitlines():
        m = re.match(r"\s+\.(name|vgpr_count|private_segment_fixed_size):\s+(\S+)", line)
        if not m:
            continue
        if m.group(1) == "name":
            cur = table.setdefault(m.group(2), {})
        else:
            cur[m.group(1)] = int(m.group(2))
    demangled = subprocess.run(["c++filt"] + list(table), check=True, capture_output=True, text=True).stdout.split("\n")
    return {d.replace("void pb::", "").replace("pb::", "").split("(")[0]: v for d, v in zip(demangled, table.values())}


# Occupancy classes ("pin occupancy classes, not only raw numbers"): waves per SIMD a kernel's VGPR count allows on
# gfx950 (512 registers per lane and SIMD, allocation granule 8: MI355X_MICROARCH.md) must not drop below what its launch bounds and
# the measurements behind them assume -- a raised VGPR budget that crosses one of these lines is an occupancy regression, not a tweak.
MIN_WAVES_PER_SIMD = {
    "k_trace<false, false, false, false>": 7, "k_trace<false, true, false, false>": 6, "k_trace<false, false, true, false>": 6,
    "k_trace<false, true, true, false>": 6, "k_trace<false, false, true, true>": 5, "k_trace<false, true, true, true>": 5,
    "k_sss_walk<false, false, true>": 3, "k_shade_principled<0>": 3, "k_shade_principled<1>": 3, "k_shade_principled<2>": 3,
    "k_tail<0, false, false, true>": 3, "k_tail<1, false, false, true>": 3, "k_shade_hair": 3,
    # k_sss_step runs at TWO waves per SIMD since round 4 (201 VGPRs with the packed light pretest): measured equal to the 156-VGPR
    # kernel (3.55 ms per 64 spp of C3 before and after, profiles/README.md round 4) -- it waits on path-state gathers, not on issue
    "k_sss_step": 2,
    "k_classify": 8, "k_compact": 5,
}


def waves_per_simd(vgprs):
    alloc = -(-vgprs // 8) * 8
    return min(8, 512 // alloc)


def test_hot_kernels_keep_their_occupancy_class():
    table = kernel_table()
    for name, waves in MIN_WAVES_PER_SIMD.items():
        assert name in table, name
        assert waves_per_simd(table[name]["vgpr_count"]) >= waves, (name, table[name], waves)


def test_hot_kernels_stay_within_their_budgets():
    table = kernel_table()
    for name, (vgprs, scratch) in BUDGETS.items():
        assert name in table, (name, sorted(table)[:8])
        got = table[name]
        assert got["vgpr_count"] <= vgprs, (name, got)
        assert got["private_segment_fixed_size"] <= scratch, (name, got)
