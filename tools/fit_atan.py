"""Coefficients of the polynomial q(s) ~ atan(t)/t, s = t^2 in [0, 1], used by the demodulation stage's atan2
(kernels_post8.hip: atan2_pi).  Near-minimax: interpolation at Chebyshev nodes in long double, converted to powers of s in
exact rational arithmetic, rounded once to double.  Prints the coefficients and the error of the double Horner evaluation."""
import sys
from fractions import Fraction

import numpy as np

DEG = int(sys.argv[1]) if len(sys.argv) > 1 else 16
LD = np.longdouble
n = DEG + 1
k = np.arange(n, dtype=LD)
u = np.cos(np.pi * (k + LD(0.5)) / n)                   # nodes in [-1, 1]
s = (u + 1) / 2
t = np.sqrt(s)
f = np.arctan(t) / t
# Chebyshev coefficients by the discrete orthogonality relation
c = np.array([(2 / LD(n)) * np.sum(f * np.cos(np.pi * j * (k + LD(0.5)) / n)) for j in range(n)], dtype=LD)
c[0] /= 2
# T_j(u) in powers of u (exact integers), u = 2 s - 1 in powers of s (exact)
T = [[Fraction(1)], [Fraction(0), Fraction(1)]]
for j in range(2, n):
    a = [Fraction(0)] + [2 * v for v in T[j - 1]]
    b = T[j - 2] + [Fraction(0)] * (len(a) - len(T[j - 2]))
    T.append([x - y for x, y in zip(a, b)])
pu = [Fraction(0)] * n
for j in range(n):
    cj = Fraction(*LD(c[j]).as_integer_ratio()) if hasattr(LD(c[j]), 'as_integer_ratio') else Fraction(float(c[j]))
    for i, v in enumerate(T[j]):
        pu[i] += cj * v
# substitute u = 2 s - 1
ps = [Fraction(0)] * n
pw = [Fraction(1)]                                      # (2 s - 1)^i in powers of s
for i in range(n):
    for d, v in enumerate(pw):
        ps[d] += pu[i] * v
    nxt = [Fraction(0)] * (len(pw) + 1)
    for d, v in enumerate(pw):
        nxt[d] -= v
        nxt[d + 1] += 2 * v
    pw = nxt
coef = [float(v) for v in ps]
print('deg', DEG)
print(', '.join(f'{v!r}' for v in coef))
# error of double Horner against long double atan over a dense grid
tt = np.linspace(0, 1, 400001)
ss = tt * tt
acc = np.full_like(ss, coef[-1])
for v in coef[-2::-1]:
    acc = acc * ss + v
approx = acc * tt
ref = np.arctan(tt.astype(LD))
print('max abs err atan(t), t in [0,1]:', float(np.abs(approx.astype(LD) - ref).max()))
