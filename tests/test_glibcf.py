"""CPU: include/pbr_glibcf.h -- the restatement of GNU libc's cosf / sinf / expf / logf (the reference calls std::cos ... on float,
i.e. these, wherever its libm is glibc) that the HIP kernels compute with since round 5 -- against the HOST's libm.

Pin status: on Ubuntu 22.04's glibc 2.35 / x86-64 / FMA (this container, the GPU boxes) every one of the 2^32 arguments of each of the four
functions gives the same bits, and the test below checks exactly that: all 4 x 2^32 (20 s on 8 threads).
On a host whose libm is another one the comparison is reported and the test SKIPS: the functions are then still what the GPU
computes (GPU == oracle[glibcf] is asserted by the -m gpu tests on any host), only "equal to the reference's own arithmetic" is
about another platform."""
import os

import numpy as np
import pytest

import _oracle as O


def test_glibcf_equals_the_host_libm():
    """ALL 2^32 bit patterns of each of the four functions (17.2 G comparisons, ~20 s on 8 threads; PBR_GLIBCF_QUICK=1 samples every
    61st instead).  On the image this project is built and run on (glibc 2.35, x86-64) any difference FAILS; elsewhere it skips."""
    import platform
    L = O.lib()
    if os.environ.get("PBR_GLIBCF_QUICK"):
        stride, bad = 61, int(L.orc_glibcf_vs_libm(61, 0))
    else:  # interleaved over the host's threads (ctypes releases the GIL)
        from concurrent.futures import ThreadPoolExecutor
        stride, n = 1, O.oracle_threads()
        with ThreadPoolExecutor(n) as ex:
            bad = sum(ex.map(lambda k: int(L.orc_glibcf_vs_libm(n, k)), range(n)))
    here = (platform.libc_ver(), platform.machine())
    if bad and here != (("glibc", "2.35"), "x86_64"):
        pytest.skip(f"the host libm is not the one include/pbr_glibcf.h restates ({bad} of {4 * (2**32 // stride)} results differ; {here}): "
                    f"parity with oracle[libm] is a tolerance here, not bits")
    assert bad == 0, (bad, here)
    print(f"glibcf == libm on {4 * (2**32 // stride)} arguments (stride {stride})")


def test_oracle_modes_agree_when_the_libm_is_glibc():
    """the checker rendered with the host libm and with the restatement: the same image, bit for bit"""
    if not O.libm_is_glibcf():
        pytest.skip("host libm differs from include/pbr_glibcf.h")
    from pbrlab_amd import scenes
    so = O.oracle_scene_from_desc(scenes.cornell_scene("sss", monkey_subdiv=2, lucy_nu=64, lucy_nv=12))
    a, ca, _ = so.render(96, 64, 8, threads=4, math_mode=O.MATH_LIBM)
    b, cb, _ = so.render(96, 64, 8, threads=4, math_mode=O.MATH_GLIBCF)
    assert np.array_equal(ca, cb) and a.tobytes() == b.tobytes()
