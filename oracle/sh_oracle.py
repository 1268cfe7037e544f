"""Real spherical harmonics, CPU restatement of shencoder/src/shencoder.cu:50-355 (float64, numpy).

TEST INFRASTRUCTURE ONLY -- never imported by the product package.

The reference writes one polynomial per output and per partial derivative.  This restatement derives the same
polynomials independently of the HIP kernel's recurrences, from numpy's Legendre machinery:

    Y[l*l + l + m] = N(l,|m|) * (d^|m| P_l / dz^|m|)(z) * Re/Im (x + i y)^|m|

which is exactly the family the reference tabulates (x, y, z independent, unit-sphere form in z).  It is pinned
against the reference by `REFERENCE_SPOT_TERMS`: a few of the reference's own polynomials, each cited by line,
re-typed as python lambdas and checked in tests/test_oracle_sh.py.  (Parity otherwise unpinned: the reference has
no SH test vectors.)
"""
from math import factorial, pi, sqrt

import numpy as np
from numpy.polynomial import legendre as _leg


def _norm(l, m):
    v = sqrt((1.0 if m == 0 else 2.0) * (2 * l + 1) / (4 * pi) * factorial(l - m) / factorial(l + m))
    return -v if (m & 1) else v


def _q(l, m, z, extra=0):
    """(d^(m+extra)/dz^(m+extra)) P_l evaluated at z."""
    k = m + extra
    if k > l:
        return np.zeros_like(z)
    return _leg.Legendre.basis(l).deriv(k)(z) if k > 0 else _leg.Legendre.basis(l)(z)


def sh_encode(inputs, degree, calc_grad=False):
    """inputs [B,3] -> outputs [B, degree^2] float64 (and dy_dx [B, 3, degree^2] laid out as the reference: d-major)."""
    v = np.asarray(inputs, dtype=np.float64)
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    B = v.shape[0]
    C2 = degree * degree
    w = x + 1j * y
    pw = [np.ones(B, dtype=np.complex128)]
    for _ in range(1, degree + 1):
        pw.append(pw[-1] * w)
    out = np.zeros((B, C2))
    jac = np.zeros((B, 3, C2)) if calc_grad else None
    for l in range(degree):
        for m in range(0, l + 1):
            n = _norm(l, m)
            q = _q(l, m, z)
            a, b = pw[m].real, pw[m].imag
            ip, im = l * l + l + m, l * l + l - m
            out[:, ip] = n * q * a
            if m > 0:
                out[:, im] = n * q * b
            if calc_grad:
                qz = _q(l, m, z, 1)
                jac[:, 2, ip] = n * qz * a
                if m > 0:
                    a1, b1 = pw[m - 1].real, pw[m - 1].imag
                    jac[:, 2, im] = n * qz * b
                    jac[:, 0, ip] = n * q * m * a1
                    jac[:, 1, ip] = -n * q * m * b1
                    jac[:, 0, im] = n * q * m * b1
                    jac[:, 1, im] = n * q * m * a1
    if calc_grad:
        return out, jac.reshape(B, 3 * C2)
    return out


def sh_encode_backward(grad, dy_dx, degree):
    """shencoder.cu:359-383: grad_inputs[b,d] = sum_ch grad[b,ch] * dy_dx[b,d,ch]."""
    B = grad.shape[0]
    C2 = degree * degree
    return np.einsum("bc,bdc->bd", np.asarray(grad, np.float64), np.asarray(dy_dx, np.float64).reshape(B, 3, C2))


# A sample of the reference's own polynomials: (output index, shencoder.cu line, f(x, y, z)).
REFERENCE_SPOT_TERMS = [
    (0, 51, lambda x, y, z: 0.28209479177387814 + 0 * x),
    (1, 53, lambda x, y, z: -0.48860251190291987 * y),
    (2, 54, lambda x, y, z: 0.48860251190291987 * z),
    (3, 55, lambda x, y, z: -0.48860251190291987 * x),
    (6, 59, lambda x, y, z: 0.94617469575755997 * z * z - 0.31539156525251999),
    (8, 61, lambda x, y, z: 0.54627421529603959 * x * x - 0.54627421529603959 * y * y),
    (9, 63, lambda x, y, z: 0.59004358992664352 * y * (-3.0 * x * x + y * y)),
    (12, 66, lambda x, y, z: 0.3731763325901154 * z * (5.0 * z * z - 3.0)),
    (15, 69, lambda x, y, z: 0.59004358992664352 * x * (-x * x + 3.0 * y * y)),
    (20, 75, lambda x, y, z: -3.1735664074561294 * z**2 + 3.7024941420321507 * z**4 + 0.31735664074561293),
    (27, 83, lambda x, y, z: -0.48923829943525038 * y * (3.0 * x * x - y * y) * (9.0 * z * z - 1.0)),
    (42, 99, lambda x, y, z: 6.6747662381009842 * z**2 - 20.024298714302954 * z**4 + 14.684485723822165 * z**6 - 0.31784601133814211),
    (63, 121, lambda x, y, z: 0.70716273252459627 * x * (-35.0 * x**2 * y**4 + 21.0 * x**4 * y**2 - x**6 + 7.0 * y**6)),
]
