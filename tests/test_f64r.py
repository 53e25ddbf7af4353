"""CPU: the shared transcendental functions (include/pbr_f64r.h, compiled verbatim by the HIP kernels and by the checker's
ORC_MATH_F64R mode) against the host's double-precision functions: 10^7 samples per function, error at most
0.5 ulp_float + 2^-20 ulp_float -- i.e. the double result before the rounding is within 2^-44 relative of the true value."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O

N = 10_000_000


def f64r(op, x):
    L = O.lib()
    L.orc_kat_f64r.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    L.orc_kat_f64r.restype = None
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    L.orc_kat_f64r(op, x.ctypes.data, x.size, out.ctypes.data)
    return out


def ulp_of(ref):
    """spacing of the floats around |ref| (float64 array), denormal range included"""
    a = np.abs(ref)
    e = np.floor(np.log2(np.maximum(a, 2.0 ** -126)))
    return 2.0 ** (np.maximum(e, -126) - 23)


def check(got, ref, what):
    ok = np.isfinite(ref)
    err = np.abs(got[ok].astype(np.float64) - ref[ok]) / ulp_of(ref[ok])
    worst = float(err.max())
    assert worst <= 0.5 + 2.0 ** -20, (what, worst, int(err.argmax()))
    return worst


def samples(seed):
    rng = np.random.RandomState(seed)
    return rng


@pytest.mark.parametrize("op,name", [(0, "sin"), (1, "cos")])
def test_sin_cos(op, name):
    rng = np.random.RandomState(10 + op)
    # the callers' range [0, 2 pi] (2 pi u with u = Draw(): 23-bit mantissa), a wider band, and the neighbourhood of multiples of pi / 2
    x = np.concatenate([
        (np.float32(2 * np.pi) * rng.rand(N // 2).astype(np.float32)),
        (rng.rand(N // 4).astype(np.float32) * np.float32(200.0) - np.float32(100.0)),
        (np.round(rng.rand(N // 4) * 64).astype(np.float32) * np.float32(np.pi / 2) * (1 + (rng.rand(N // 4).astype(np.float32) - 0.5) * np.float32(1e-5))),
    ]).astype(np.float32)
    ref = (np.sin if op == 0 else np.cos)(x.astype(np.float64))
    worst = check(f64r(op, x), ref, name)
    print(name, "worst error", worst, "ulp")
    # edge cases: sin 0 = 0 and cos 0 = 1 exactly (the sign of a negative zero is not kept: the callers pass 2 pi u >= +0);
    # non-finite arguments give NaN
    z = f64r(op, np.array([0.0, -0.0, np.inf, -np.inf, np.nan], np.float32))
    assert np.isnan(z[2:]).all() and (z[:2] == (0.0 if op == 0 else 1.0)).all()


def test_exp():
    rng = np.random.RandomState(20)
    x = np.concatenate([
        -rng.rand(N // 2).astype(np.float32) * np.float32(12.0),                 # the walk's transmittances and Burley's fit
        (rng.rand(N // 4).astype(np.float32) * np.float32(195.0) - np.float32(106.0)),   # the whole float range of the result, denormals included
        -np.exp(rng.rand(N // 4) * 40 - 30).astype(np.float32),                  # tiny and huge negative arguments
    ]).astype(np.float32)
    with np.errstate(over="ignore", under="ignore"):
        ref = np.exp(x.astype(np.float64))
    got = f64r(2, x)
    fin = ref < 3.4028234663852886e38
    worst = check(got[fin], ref[fin], "exp")
    print("exp worst error", worst, "ulp")
    e = f64r(2, np.array([0.0, -np.inf, np.inf, np.nan, -200.0, 89.0, 88.0, -103.0, -104.0], np.float32))
    assert e[0] == 1.0 and e[1] == 0.0 and np.isinf(e[2]) and np.isnan(e[3]) and e[4] == 0.0 and np.isinf(e[5])
    assert e[6] == np.float32(np.exp(88.0)) and e[7] == np.float32(np.exp(-103.0)) and e[8] == np.float32(np.exp(np.float64(-104.0)))


def test_log():
    rng = np.random.RandomState(30)
    u = rng.randint(0, 1 << 23, size=N // 2).astype(np.float32) * np.float32(2.0 ** -23)
    x = np.concatenate([
        np.float32(1.0) - u,                                                      # log(1 - Draw()): (0, 1]
        np.exp(rng.rand(N // 4) * 170 - 100).astype(np.float32),                  # the float range, denormals included
        (np.float32(1.0) + (rng.rand(N // 4).astype(np.float32) - np.float32(0.5)) * np.float32(1e-3)),   # around 1
    ]).astype(np.float32)
    x = x[x > 0]
    ref = np.log(x.astype(np.float64))
    got = f64r(3, x)
    nz = ref != 0
    worst = check(got[nz], ref[nz], "log")
    assert (got[~nz] == 0).all()
    print("log worst error", worst, "ulp")
    e = f64r(3, np.array([1.0, 0.0, -1.0, np.inf, np.nan, 1e-45], np.float32))
    assert e[0] == 0.0 and e[1] == -np.inf and np.isnan(e[2]) and np.isinf(e[3]) and np.isnan(e[4])
    assert e[5] == np.float32(np.log(np.float64(np.float32(1e-45))))
