"""Adams-Bashforth / Adams-Moulton coefficients, generated exactly from their definition.

bashforth(k)[j] multiplies f_{n-j} (j = 0..k-1) in the k-step explicit formula
    y_{n+1} = y_n + h * sum_j b_j f_{n-j},        b_j = int_0^1 prod_{i != j} (s + i) / (i - j) ds,
moulton(k)[j] multiplies f_{n+1-j} in the k-term implicit formula
    y_{n+1} = y_n + h * sum_j m_j f_{n+1-j},      m_j = int_0^1 prod_{i != j} (s - 1 + i) / (i - j) ds
(the integrals of the Lagrange basis polynomials through the last k derivative samples).  The reference stores the
same numbers as integer tables with a common divisor (paddlexde/solver/fixed_solver/adams.py:9-438); e.g.
bashforth(4) = [55, -59, 37, -9] / 24 and moulton(4) = [9, 19, -5, 1] / 24.
"""
from fractions import Fraction
from functools import lru_cache


def _poly_mul(p, q):
    out = [Fraction(0)] * (len(p) + len(q) - 1)
    for i, a in enumerate(p):
        for j, b in enumerate(q):
            out[i + j] += a * b
    return out


def _integrate_01(p):
    return sum(c / (i + 1) for i, c in enumerate(p))


def _basis_integrals(nodes):
    """int_0^1 of each Lagrange basis polynomial for the given nodes."""
    out = []
    for j, xj in enumerate(nodes):
        poly = [Fraction(1)]
        for i, xi in enumerate(nodes):
            if i != j:
                poly = _poly_mul(poly, [Fraction(-xi) / (xj - xi), Fraction(1) / (xj - xi)])
        out.append(_integrate_01(poly))
    return out


@lru_cache(maxsize=None)
def bashforth(k):
    """Coefficients of f_n, f_{n-1}, ..., f_{n-k+1} (nodes s = 0, -1, ..., -(k-1))."""
    return tuple(_basis_integrals([Fraction(-i) for i in range(k)])) if k > 0 else ()


@lru_cache(maxsize=None)
def moulton(k):
    """Coefficients of f_{n+1}, f_n, ..., f_{n-k+2} (nodes s = 1, 0, ..., -(k-2))."""
    return tuple(_basis_integrals([Fraction(1 - i) for i in range(k)])) if k > 0 else ()
