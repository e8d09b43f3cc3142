"""Constants of the table-driven device exp2: T[j] = 2**(j/32), j = 0..31, and the degree-5 near-minimax
polynomial of 2**(z/32) on z in [-1/2, 1/2] (coefficients already divided by 32**i).
Run: python tools/gen_exp2_table.py  -> C initialisers for pymbar_amd/csrc/mbar_kernels.hip"""
import numpy as np

LD = np.longdouble
DEG = 6
n = DEG + 1
pi = LD("3.14159265358979323846264338327950288")
j = np.arange(n, dtype=LD)
nodes = np.cos((j + LD(0.5)) * pi / n)
z = nodes * LD(0.5)                         # z in [-1/2, 1/2]
f = np.exp2(z / LD(32))
c = np.array([(LD(2) / n) * np.sum(f * np.cos(k * (j + LD(0.5)) * pi / n)) for k in range(n)], dtype=LD)
c[0] /= 2
T = [np.zeros(n, dtype=LD) for _ in range(n)]
T[0][0] = 1
T[1][1] = 1
for k in range(1, n - 1):
    T[k + 1][1:] = 2 * T[k][:-1]
    T[k + 1] -= T[k - 1]
mono_y = sum(c[k] * T[k] for k in range(n))  # in y = 2 z
mono_z = mono_y * (LD(2) ** np.arange(n, dtype=LD))
coef = np.array([float(v) for v in mono_z])
zz = np.linspace(LD(-0.5), LD(0.5), 200001, dtype=LD)
p = np.zeros_like(zz)
for v in coef[::-1]:
    p = p * zz + LD(v)
print("// max relative error of the rounded degree-%d polynomial: %.3g" % (DEG, float(np.max(np.abs(p / np.exp2(zz / 32) - 1)))))
print("// polynomial coefficients c0..c%d of 2^(z/32):" % DEG)
print(", ".join(float(v).hex() for v in coef))
print("// table 2^(j/32):")
tab = [float(np.exp2(LD(jj) / LD(32))) for jj in range(32)]
for i in range(0, 32, 4):
    print("    " + ", ".join(v.hex() for v in tab[i:i + 4]) + ",")
