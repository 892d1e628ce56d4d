#!/usr/bin/env python
"""Coefficients of the polynomial GELU used by the GEMM epilogues (vitpose_amd/csrc/common.h).

gelu(x)  = (x + |x|) / 2 - h(|x|),          h(a) = a erfc(a / sqrt 2) / 2
gelu'(x) = 1/2 + copysign(1/2 - d(|x|), x), d(a) = h'(a) = erfc(a / sqrt 2) / 2 - a phi(a)
h and d are smooth and decay like a Gaussian; on [0, A] each is fitted by a degree-DEG polynomial in
s = 2 a / A - 1 (Chebyshev least squares on a dense grid, converted to the monomial basis, whose
coefficients stay below 2 in magnitude on [-1, 1]), and a is clamped to A beyond.  The script prints the
coefficients and the maximum absolute error of the float32 Horner evaluation against scipy's erfc."""
import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import erfc

A, DEG = 4.5, 11


def h(a):
    return a * erfc(a / np.sqrt(2)) / 2


def d(a):
    return erfc(a / np.sqrt(2)) / 2 - a * np.exp(-a * a / 2) / np.sqrt(2 * np.pi)


def horner32(mono, s):
    acc = np.full_like(s, np.float32(mono[-1]))
    for k in mono[-2::-1]:
        acc = acc * s + np.float32(k)
    return acc


def main():
    a = np.linspace(0, A, 20001)
    s = 2 * a / A - 1
    x = np.linspace(-9, 9, 400001).astype(np.float32)
    ax = np.minimum(np.abs(x), np.float32(A))
    sx = ax * np.float32(2 / A) - np.float32(1)
    for name, f in (('H', h), ('D', d)):
        mono = C.cheb2poly(C.chebfit(s, f(a), DEG))
        print(f'// {name}: degree {DEG} in s = 2 a / {A} - 1, highest power first')
        print('{' + ', '.join(f'{c:.9e}f' for c in mono[::-1]) + '}')
        p = horner32(mono, sx).astype(np.float64)
        xd = x.astype(np.float64)
        if name == 'H':
            got = 0.5 * (xd + np.abs(xd)) - p
            ref = xd * (1 - erfc(xd / np.sqrt(2)) / 2)
        else:
            got = 0.5 + np.copysign(0.5 - p, xd)
            ref = (1 - erfc(xd / np.sqrt(2)) / 2) + xd * np.exp(-xd * xd / 2) / np.sqrt(2 * np.pi)
        print(f'// max |error| on [-9, 9]: {np.abs(got - ref).max():.2e}')


if __name__ == '__main__':
    main()
