"""The branch-free sine/cosine of the device rollouts (ilqr_amd/csrc/fast_trig.hpp) compiled for the
host with g++ and compared with long-double libm: the header is the one the kernels include, so this
checks the code that runs on the GPU (same operations, fused multiply-adds, no contraction)."""
import os
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROGRAM = r"""
#include <cmath>
#include <cstdio>
#include <cstdint>
#include "fast_trig.hpp"
static double ulps(double got, long double want) {
    if (want == 0.0L) return got == 0.0 ? 0.0 : 1e9;
    int e;
    std::frexp((double)want, &e);
    return (double)(fabsl((long double)got - want) / ldexpl(1.0L, e - 53));
}
int main() {
    double worst = 0.0, at = 0.0;
    uint64_t st = 88172645463325252ull;
    for (long i = 0; i < 4000000; ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        const double u = (double)(st >> 11) / 9007199254740992.0;
        const double range = (i % 4 == 0) ? 8.0 : (i % 4 == 1) ? 1000.0 : (i % 4 == 2) ? 1048575.0 : 0.8;
        const double x = (2.0 * u - 1.0) * range;
        double s, c;
        if (!ilqr::fast_sincos(x, s, c)) { std::printf("range %.17g\n", x); return 1; }
        const double e = std::fmax(ulps(s, sinl((long double)x)), ulps(c, cosl((long double)x)));
        if (e > worst) { worst = e; at = x; }
    }
    double s, c;
    const double out[] = {1048576.0, -1048576.0, 1e300, INFINITY, -INFINITY, NAN};
    for (double x : out)
        if (ilqr::fast_sincos(x, s, c)) { std::printf("accepted %.17g\n", x); return 1; }
    // exact values at the quadrant boundaries' neighbours that are representable
    if (!ilqr::fast_sincos(0.0, s, c) || s != 0.0 || c != 1.0) { std::printf("zero\n"); return 1; }
    std::printf("worst %.4f ulp at %.17g\n", worst, at);
    return 0;
}
"""


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_fast_sincos_against_long_double_libm():
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "trig.cpp")
        with open(src, "w") as f:
            f.write(PROGRAM)
        exe = os.path.join(d, "trig")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "ilqr_amd", "csrc"),
                               "-o", exe, src])
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout
        worst = float(out.stdout.split()[1])
        assert worst <= 1.6, out.stdout
