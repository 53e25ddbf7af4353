"""CPU: the parts of bench.py that need no GPU -- the ISA-histogram VALU model (round 6: the ceiling `roofline.valu.frac` is SQ_INSTS_VALU x the mean
measured cost of the instructions of the kernel's main loop, read from the gfx950 code object inside the built libpbrhip.so) and the metric's
bookkeeping."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (importing it touches neither torch nor the GPU)


def test_valu_cost_classes():
    # full rate (2.33 cycles measured: scripts/ubench/valu_rate*.hip): plain fp32 add / mul / fma, simple integer, moves, selects
    for m in ("v_fma_f32", "v_fmac_f32_e32", "v_add_f32_e32", "v_mul_f32_e64", "v_and_b32_e32", "v_mov_b32_e32", "v_cndmask_b32_e64", "v_add_u32_e32"):
        assert bench.valu_cost(m) == bench.VALU_FULL, m
    # half rate: packed fp32, min / max, compares, conversions, shifts, integer multiplies
    for m in ("v_pk_fma_f32", "v_pk_mul_f32", "v_max_f32_e32", "v_min3_f32", "v_cmp_le_f32_e32", "v_cvt_f32_ubyte1_e32", "v_lshlrev_b32_e32", "v_mul_lo_u32", "v_bfe_u32"):
        assert bench.valu_cost(m) == bench.VALU_HALF, m
    for m in ("v_rcp_f32_e32", "v_sqrt_f32_e32", "v_rsq_f32_e32", "v_exp_f32_e32"):
        assert bench.valu_cost(m) == bench.VALU_TRANS, m
    assert bench.valu_cost("v_fma_f64") == bench.VALU_F64


def test_isa_histogram_of_the_built_kernels():
    if not (os.path.exists(os.path.join(ROOT, "pbrlab_amd", "libpbrhip.so")) and os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump") and shutil.which("c++filt")):
        pytest.skip("built library or LLVM tools not available")
    for kernel in ("k_trace<false, false, true, false>", "k_trace<false, true, true, false>", "k_shade_principled<0>"):
        h = bench.isa_histogram(kernel)
        assert h is not None and h["kernel"].startswith(kernel.split("<")[0]), kernel
        # the main loop of a traversal kernel: hundreds of vector instructions, a few dozen vector-memory ones, a mean cost between the two rates
        assert h["valu_in_loop"] > 300 and h["vmem_in_loop"] >= 4 and h["salu_in_loop"] > 50, h
        assert bench.VALU_FULL < h["mean_cycles_per_valu"] < bench.VALU_HALF + 0.3, h
        assert 0.2 < h["full_rate_share"] < 0.8, h
    assert bench.isa_histogram("no_such_kernel") is None


def test_workloads_are_the_baseline_configurations():
    w = bench.WORKLOADS
    assert (w["c2"]["width"], w["c2"]["height"], w["c2"]["spp"]) == (1920, 1080, 64)       # BASELINE configs[1]: the metric's configuration
    assert (w["c3"]["spp"], w["c4"]["spp"]) == (256, 128) and (w["c5"]["width"], w["c5"]["height"], w["c5"]["spp"]) == (3840, 2160, 1024)
    assert bench.HBM_PEAK_GBS == 8000.0
