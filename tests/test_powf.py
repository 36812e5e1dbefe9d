"""powf of phongShade (flyscene.cpp:852): the oracle's restatement of glibc 2.35's e_powf.c (FMA build) against the host's libm powf.

Exhaustive over every float in [0, 1 + 2^-10] whose power can be non-zero, for the shininess exponents of every shipped .mtl
(10, 225, 324, 500), plus strided sweeps with other exponents.  The comparison only means something on a host whose libm ifunc
selects the FMA build (x86-64 with FMA + AVX2: this container, the GPU box and the machine the reference md5s were made on);
elsewhere libm's own results differ in rare last-place cases from the reference's and the test is skipped with that message.
"""
import ctypes as C

import pytest


def _cpu_has_fma():
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        return False
    return " fma " in flags and " avx2 " in flags


@pytest.fixture(scope="module")
def cmp(oracle):
    f = oracle.lib.orc_powf_compare
    f.restype = C.c_long
    f.argtypes = [C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    return f


@pytest.mark.skipif(not _cpu_has_fma(), reason="host libm uses the non-FMA powf build: its last-place results are not the reference's")
@pytest.mark.parametrize("expo,lo", [(10.0, 0x38000000), (225.0, 0x3f000000), (324.0, 0x3f000000), (500.0, 0x3f000000)])
def test_orc_powf_is_the_hosts_powf_exhaustively(cmp, expo, lo):
    """every base from `lo` (below it the power underflows to zero -- covered by the strided sweep) up to 1 + 2^-10"""
    bad = C.c_uint32()
    n = cmp(expo, lo, 0x3f802000, 1, C.byref(bad))
    assert n == 0, f"{n} results differ from libm, first at base bits {bad.value:#x}"


@pytest.mark.skipif(not _cpu_has_fma(), reason="host libm uses the non-FMA powf build")
@pytest.mark.parametrize("expo", [10.0, 225.0, 324.0, 500.0, 0.5, 1.0, 2.0, 17.3, 64.0, 1000.0])
def test_orc_powf_strided_over_all_bases_up_to_one(cmp, expo):
    """zero, subnormal and normal bases up to 1 + 2^-7, every 61st bit pattern"""
    bad = C.c_uint32()
    n = cmp(expo, 0, 0x3f810000, 61, C.byref(bad))
    assert n == 0, f"{n} results differ from libm, first at base bits {bad.value:#x}"
