"""Shared check of the GPU Metropolis tests: walkers whose chain differs from the oracle's must have taken a
rounding-borderline decision.

The kernels sum in another order than the oracle (K split over lanes, MFMA accumulation), so a proposal whose Jacobian
ratio sits within float32 rounding of `u`, whose likelihood sits within rounding of L*, or whose coordinate sits within
rounding of the box edge can be decided the other way; from that step on the two chains are different chains.  The tests
exclude such walkers from the element-wise comparison -- and assert here that the exclusion is what it claims to be: at the
FIRST step where the kernel's chain leaves the oracle's, the oracle's decision margin (orc_set_margin_out,
oracle/nnest_oracle.c: min of |u - ratio|, |logL' - L*| / (1 + |L*|), ||x'_d| - 1|) is at rounding level."""
import numpy as np

# float32: log-det sums of ~75 terms of magnitude <~ 50 differ by a few ulp (~1e-5) between summation orders, the ratio
# e^dlogdet inherits that relative error; the likelihood is a float64 sum of float32 terms of a float32 x' that itself
# differs by ~1e-6 relative -- on Rosenbrock's ridge (100 (x_{i+1} - x_i^2)^2) that is ~1e-5 of logL
MARGIN_TOL = 1e-4


def first_divergence(h_gpu, h_orc, tol=1e-3):
    """per walker: first step index s >= 1 where the chains differ by more than tol (relative), or -1"""
    d = np.max(np.abs(h_gpu - h_orc) / (1.0 + np.abs(h_orc)), axis=2) > tol      # [C, S+1]
    first = np.where(d.any(axis=1), d.argmax(axis=1), -1)
    return first


def assert_borderline(h_gpu, h_orc, margins, walkers, tol=MARGIN_TOL):
    """walkers: indices excluded from the element-wise comparison.  Each must leave the oracle's chain at a step the oracle
    decided within `tol` of a threshold."""
    first = first_divergence(h_gpu[walkers], h_orc[walkers])
    for w, s in zip(np.asarray(walkers), first):
        assert s >= 1, 'walker %d was excluded but its chain equals the oracle\'s' % w
        assert margins[s - 1, w] < tol, ('walker %d leaves the oracle chain at step %d where the oracle\'s decision margin is %.3g '
                                         '(not a rounding-borderline decision)' % (w, s, margins[s - 1, w]))
