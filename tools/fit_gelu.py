"""The sigmoid-form fit of GELU used by the GEMM epilogues (csrc/wg_common.h wg_act2<WG_ACT_GELU_ERF>): minimax fit of
x / (1 + exp(-x (a + b x^2 + c x^4))), x^2 clamped to 49, to x Phi(x) = x/2 (1 + erf(x / sqrt 2)); prints the coefficients as the kernel holds them
(times -log2 e) and the error of the fp32 evaluation the kernel performs.  CPU only (numpy, scipy)."""
import numpy as np
from scipy.optimize import minimize
from scipy.special import erf

XC = 49.0
x = np.linspace(-12, 12, 48001)
gelu = x * 0.5 * (1 + erf(x / np.sqrt(2)))


def model(p, x):
    x2 = np.minimum(x * x, XC)
    with np.errstate(over="ignore"):
        return x / (1 + np.exp(-x * (p[0] + p[1] * x2 + p[2] * x2 * x2)))


f = lambda p: np.max(np.abs(model(p, x) - gelu))
p = np.array([1.5957691, 0.07135481, 0.0])      # start: the tanh form everybody knows
for _ in range(6):
    p = minimize(f, p, method="Nelder-Mead", options=dict(xatol=1e-12, fatol=1e-14, maxiter=40000)).x
print("a, b, c =", [float(c) for c in p], " max |fit - erf form| =", f(p))
p32 = (-p * np.log2(np.e)).astype(np.float32)
xs = np.linspace(-12, 12, 480001).astype(np.float32)
x2 = np.minimum(xs * xs, np.float32(XC))
with np.errstate(over="ignore"):
    y = xs / (np.float32(1) + np.exp2(xs * (p32[0] + x2 * (p32[1] + x2 * p32[2]))))
ref = xs.astype(np.float64) * 0.5 * (1 + erf(xs.astype(np.float64) / np.sqrt(2)))
print("kernel coefficients (* -log2 e):", [float(c) for c in p32], " fp32 evaluation: max abs error", float(np.max(np.abs(y - ref))))
