"""TEST INFRASTRUCTURE (a checker; nothing under nnest_amd/ may import it).

The exact evidence of the reference's chain Rosenbrock likelihood (nnest/likelihoods.py:48-59:
log L = -sum_{i < D} [100 (x_{i+1} - x_i^2)^2 + (1 - x_i)^2]) under the uniform prior on [-a, a]^D, by transfer-operator quadrature:
the integrand is a first-order chain, so with f_D = 1 and
    f_i(x_i) = exp(-(1 - x_i)^2)  integral_{-a}^{a} exp(-100 (x_{i+1} - x_i^2)^2) f_{i+1}(x_{i+1}) dx_{i+1}
Z (2a)^D = integral f_1.  Trapezoid rule on a grid of step h (the kernel's width is 1 / sqrt(200) = 0.0707; h = 0.005 is converged to
1e-7 in log Z); O(D (2a/h)^2) flops.  It is a known answer of the likelihood, not of the sampler: no part of the reference computes it.

   python oracle/rosenbrock_exact.py [D ...]
"""
import sys
import numpy as np


def log_evidence(x_dim, half_width=5.0, h=0.005):
    x = np.arange(-half_width, half_width + h / 2, h)
    w = np.full(x.size, h)
    w[0] = w[-1] = h / 2
    kern = np.exp(-100.0 * (x[None, :] - x[:, None] ** 2) ** 2)     # [x_i, x_{i+1}]
    g = np.exp(-(1.0 - x) ** 2)
    f, log_scale = np.ones_like(x), 0.0
    for _ in range(x_dim - 1):
        f = g * (kern @ (w * f))
        m = f.max()
        f /= m
        log_scale += np.log(m)
    return float(log_scale + np.log((w * f).sum()) - x_dim * np.log(2.0 * half_width))


if __name__ == '__main__':
    for d in [int(a) for a in sys.argv[1:]] or [2, 20, 50, 100]:
        print('x_dim %3d  log Z = %.6f' % (d, log_evidence(d)))
