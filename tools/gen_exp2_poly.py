"""Derive the degree-11 polynomial used by the device exp2 core: a near-minimax (Chebyshev interpolant)
approximation of 2**r on [-0.5, 0.5] computed in 80-bit long double, rounded to fp64.  Prints C initialisers
and the estimated max relative error.  Run: python tools/gen_exp2_poly.py"""
import numpy as np

LD = np.longdouble
DEG = 11
n = DEG + 1
j = np.arange(n, dtype=LD)
pi = LD(np.pi) if False else LD("3.14159265358979323846264338327950288")
nodes = np.cos((j + LD(0.5)) * pi / n)            # Chebyshev nodes on [-1, 1]
x = nodes * LD(0.5)
f = np.exp2(x)
# Chebyshev coefficients c_k = (2/n) sum_j f(x_j) T_k(node_j)
c = np.zeros(n, dtype=LD)
for k in range(n):
    c[k] = (LD(2) / n) * np.sum(f * np.cos(k * (j + LD(0.5)) * pi / n))
c[0] /= 2
# convert sum_k c_k T_k(y), y = 2 r, to monomials in y by the recurrence T_{k+1} = 2 y T_k - T_{k-1}
T = [np.zeros(n, dtype=LD) for _ in range(n)]
T[0][0] = 1
T[1][1] = 1
for k in range(1, n - 1):
    T[k + 1][1:] = 2 * T[k][:-1]
    T[k + 1] -= T[k - 1]
mono_y = np.zeros(n, dtype=LD)
for k in range(n):
    mono_y += c[k] * T[k]
mono_r = mono_y * (LD(2) ** np.arange(n, dtype=LD))   # y = 2 r
coef = np.array([float(v) for v in mono_r])
# error estimate on a dense grid, polynomial evaluated in long double with the ROUNDED coefficients
r = np.linspace(LD(-0.5), LD(0.5), 200001, dtype=LD)
p = np.zeros_like(r)
for v in coef[::-1]:
    p = p * r + LD(v)
err = np.max(np.abs(p / np.exp2(r) - 1))
print("max relative error of the rounded polynomial (exact arithmetic):", float(err))
print("static constexpr double EXP2_C[%d] = {" % n)
for v in coef:
    print("    %s,  // %.17g" % (float(v).hex(), v))
print("};")
