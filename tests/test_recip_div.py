"""The kernels divide by constants (3.6) through q = a*r; q + fma(-q, b, a)*r with r = RN(1/b) (csrc/fcpp_pointfn.h div_by): this
checks, with the host's fma (C99 fma() is exact), that the three instructions give the correctly rounded quotient -- i.e. the same
bits as the IEEE division the oracle and the reference perform -- on 4*10^7 random operands and on the speeds a plan can hold."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

SRC = r'''
#include <math.h>
#include <stdint.h>
static uint64_t s;
static inline uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
long check(double b, long n, uint64_t seed, int emin, int espan)
{
    const double r = 1.0 / b;
    long bad = 0;
    s = seed;
    for (long i = 0; i < n; ++i) {
        const uint64_t u = rnd();
        union { uint64_t u; double d; } c;
        c.u = (u & 0x000fffffffffffffull) | ((uint64_t)(1023 + emin + (int)((u >> 52) % (unsigned)espan)) << 52);
        const double a = c.d, q = a * r, q2 = fma(fma(-q, b, a), r, q);
        bad += q2 != a / b;
    }
    return bad;
}
long check_list(double b, long n, const double *a)
{
    const double r = 1.0 / b;
    long bad = 0;
    for (long i = 0; i < n; ++i) { const double q = a[i] * r; bad += fma(fma(-q, b, a[i]), r, q) != a[i] / b; }
    return bad;
}
'''


def test_reciprocal_division_is_exact():
    with tempfile.TemporaryDirectory() as d:
        src, lib = os.path.join(d, 'rd.c'), os.path.join(d, 'librd.so')
        open(src, 'w').write(SRC)
        subprocess.check_call(['gcc', '-O2', '-ffp-contract=off', '-shared', '-fPIC', '-o', lib, src, '-lm'])
        L = C.CDLL(lib)
        L.check.restype = C.c_long
        L.check.argtypes = [C.c_double, C.c_long, C.c_uint64, C.c_int, C.c_int]
        L.check_list.restype = C.c_long
        L.check_list.argtypes = [C.c_double, C.c_long, C.POINTER(C.c_double)]
        for b in (3.6, 9.0 / 3.6, 15.0 / 3.6, 4.0 / 3.6, 2.5 / 3.6, 0.1):
            assert L.check(b, 10_000_000 if b == 3.6 else 6_000_000, 88172645463325252, -30, 40) == 0, b
        # every speed a plan holds lies in (0, 200] km/h: a dense sweep of that range as well
        a = np.ascontiguousarray(np.concatenate([np.linspace(1e-3, 200.0, 2_000_001), np.random.default_rng(5).uniform(0, 60, 2_000_000)]))
        assert L.check_list(3.6, len(a), a.ctypes.data_as(C.POINTER(C.c_double))) == 0
