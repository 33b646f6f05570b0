"""Fit of the odd polynomial used by the bf16 epilogues' GELU (uc2_amd/csrc/common.h, phi_bf): logit(Phi(x)) ~= x*(c0 + c1 x^2 + c2 x^4),
iteratively re-weighted least squares towards the minimax error of x*Phi(x).  Not a test; prints the coefficients."""
import warnings
import numpy as np
from scipy.optimize import least_squares
from scipy.special import ndtr

warnings.filterwarnings("ignore")
x = np.linspace(-9, 9, 20001)
Phi = ndtr(x)


def model(c, x):
    x2 = x * x
    return 1 / (1 + np.exp(-np.clip(x * (c[0] + x2 * (c[1] + x2 * c[2])), -80, 80)))


c = np.array([1.5957691, 0.071355, 0.0])
w = np.ones_like(x)
for _ in range(40):
    c = least_squares(lambda c: w * (np.abs(x) + 0.3) * (model(c, x) - Phi), c, method="lm").x
    e = np.abs((np.abs(x) + 0.3) * (model(c, x) - Phi))
    w = w * (1 + 4 * e / e.max())
    w /= w.mean()
print([float("%.9g" % v) for v in c], "max |gelu err| %.2e, max |Phi err| %.2e" % (np.abs(x * (model(c, x) - Phi)).max(), np.abs(model(c, x) - Phi).max()))
