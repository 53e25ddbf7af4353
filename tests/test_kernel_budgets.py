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
    "k_trace<false, false, false, false>": (80, 0),    # binary tree, triangles: 6 waves per SIMD (7 until round 6)
    "k_trace<false, true, false, false>": (80, 0),     # binary tree, curves: 6 waves
    "k_trace<false, false, true, false>": (80, 0),     # Q tree, triangles: 6 waves
    "k_trace<false, true, true, false>": (96, 0),      # Q tree, curves: 5 waves since round 6 (curve records: both pieces of a leaf in one turn; 6 before)
    # a group's first launch (camera rays computed in the refill path): one block per CU fewer, nothing spilled into the loop
    "k_trace<false, false, true, true>": (80, 0),      # (triangle-only scenes: the same six blocks as the other launches since round 5)
    "k_trace<false, true, true, true>": (96, 0),
    # Round 5: kernels.hip is compiled without the SLP vectoriser (csrc/Makefile) and every kernel below lost its scratch and 10-40
    # registers with it (k_tail<1, ...> 216 B -> 0, k_sss_step 155 -> 125 VGPRs, k_shade_principled<0> 133 -> 117): the budgets are
    # what the kernels compile to now, in occupancy classes (128 VGPRs = four waves per SIMD, 168 = three)
    "k_sss_walk<false, false, true>": (128, 0),
    "k_sss_walk<false, false, false>": (128, 0),
    "k_sss_walk<false, true, false>": (128, 0),
    "k_sss_walk<false, true, true>": (168, 0),       # scenes with curves: three waves per SIMD (PB_WALK_WAVES)
    "k_shade_principled<0>": (128, 0),           # no medium, no texture (C2, C4): four waves (five spill: 10.5 -> 12.7 ms on C2)
    "k_shade_principled<1>": (128, 0),           # media, no texture (C3, C5): the medium's coefficients come from the material record
    "k_shade_principled<2>": (168, 16),          # textured materials: ParamToBsdf and the medium per hit
    "k_tail<0, false, false, true>": (168, 0),   # no medium, no texture (C2)
    "k_tail<0, false, true, true>": (168, 0),    # ... with curves (C4)
    "k_tail<1, false, false, true>": (168, 0),   # media (C3)
    "k_tail<1, false, true, true>": (168, 0),    # media + curves (C5)
    "k_tail<2, false, false, true>": (168, 0),   # textured materials
    "k_trace_quad<false>": (128, 0),             # one ray per quad of lanes (small launches; off by default)
    "k_shade_hair": (128, 0),
    "k_sss_step": (128, 0),
    "k_classify": (64, 0),
    "k_compact": (104, 0),
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


# Occupancy classes (the raw budgets above were raised to whatever the code compiled to; what must not regress
# silently is the number of waves a SIMD holds): waves per SIMD = 512 // VGPRs rounded up to 8, at most 8.
MIN_WAVES_PER_SIMD = {
    "k_trace<false, false, false, false>": 6, "k_trace<false, true, false, false>": 6, "k_trace<false, false, true, false>": 6,
    "k_trace<false, true, true, false>": 5, "k_trace<false, false, true, true>": 6, "k_trace<false, true, true, true>": 5,
    "k_sss_walk<false, false, true>": 4, "k_shade_principled<0>": 4, "k_shade_principled<1>": 4, "k_shade_principled<2>": 3,
    "k_tail<0, false, false, true>": 3, "k_tail<1, false, false, true>": 3, "k_shade_hair": 4,
    "k_sss_step": 4,
    "k_classify": 8, "k_compact": 4,
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
