"""The encoder's quotients without a division instruction (quant_pair / quant_divide, jpeglibrary_amd/csrc/encode_kernels.hip):

    r0 = v_rcp_f32(d)            r  = fma(fma(-d, r0, 1), r0, r0)                      (once per table entry)
    q0 = a * r                   q1 = fma(fma(-d, q0, a), r, q0)        q = fma(fma(-d, q1, a), r, q1)

is what hipcc's IEEE division does once its scaling and fix-up steps are the identity.  The GPU tests compare streams with
the CPU restatement; this one checks the ARITHMETIC on its own, in exact rational numbers: whatever v_rcp_f32 returns within
one ulp of 1 / d (the instruction's documented accuracy -- its exact bits are not known here), the sequence ends on the
correctly rounded quotient, for every divisor a baseline quantisation table can hold, on ordinary dividends and on the ones
that sit next to a rounding tie of the quotient or of the half-to-even rounding that follows it."""
import random
from fractions import Fraction

import numpy as np


def rn32(x):
    """a Fraction rounded to the nearest float32 (ties to even), returned as a Fraction; no denormals on this path"""
    if x == 0:
        return Fraction(0)
    s = -1 if x < 0 else 1
    x = abs(x)
    e = x.numerator.bit_length() - x.denominator.bit_length()
    if Fraction(2) ** e > x:
        e -= 1
    assert Fraction(2) ** e <= x < Fraction(2) ** (e + 1) and e >= -126
    ulp = Fraction(2) ** (e - 23)
    n, rem = divmod(x, ulp)
    n = int(n)
    if rem * 2 > ulp or (rem * 2 == ulp and (n & 1)):
        n += 1
    return s * n * ulp


def fma32(a, b, c):
    return rn32(a * b + c)


def quotient(a, d, r0):
    r = fma32(fma32(-d, r0, Fraction(1)), r0, r0)
    q0 = rn32(a * r)
    q1 = fma32(fma32(-d, q0, a), r, q0)
    return fma32(fma32(-d, q1, a), r, q1)


def f32(v):
    return Fraction(float(np.float32(v)))


def test_the_division_free_quotient_is_the_correctly_rounded_one():
    rng = random.Random(7)
    checked = 0
    for d_int in range(1, 256):
        d = Fraction(d_int)
        exact_r = rn32(1 / d)
        r_f32 = np.float32(float(exact_r))
        assert Fraction(float(r_f32)) == exact_r
        r_up = Fraction(float(np.nextafter(r_f32, np.float32(2)))), Fraction(float(np.nextafter(r_f32, np.float32(0))))
        dividends = [f32(rng.uniform(-32768, 32768)) for _ in range(10)]
        dividends += [f32(rng.uniform(-4, 4)) for _ in range(4)] + [f32(1e-5), f32(-3e-6), Fraction(0), f32(32767.996)]
        for _ in range(8):  # next to k + 0.5: the rounding behind the division flips on the last bit
            k = rng.randrange(0, 2048)
            t = np.float32(float(d_int * (k + 0.5)))
            dividends += [Fraction(float(t)), Fraction(float(np.nextafter(t, np.float32(np.inf)))), Fraction(float(np.nextafter(t, np.float32(-np.inf))))]
        for r0 in (exact_r, r_up[0], r_up[1]):
            for a in dividends:
                assert quotient(a, d, r0) == rn32(a / d), (d_int, float(a), float(r0))
                checked += 1
    assert checked > 30000
