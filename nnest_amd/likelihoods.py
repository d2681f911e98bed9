"""Analytic test likelihoods with the reference's host protocol (nnest/likelihoods.py:7-22):
`like(x[N,D] or x[D]) -> logl`, `.x_dim`, `.num_evaluations`.  The host evaluation is vectorised numpy in
the dtype of the input (it is what the reference's Python row loop computes, used for the initial live
points, nested.py:228).  `hip_like_id` tells the sampler that the same function exists inside the fused
HIP kernels (include/nnest_hip.h NNEST_LIKE_*), which is where the MCMC evaluations happen."""
import numpy as np
import scipy.special

from ._lib import LIKE_IDS


class Likelihood(object):
    num_derived = 0
    hip_like_id = None
    hip_like_params = ()

    def __init__(self, x_dim):
        self.x_dim = x_dim
        self.num_evaluations = 0

    def __call__(self, x):
        if isinstance(x, list):
            x = np.array(x)
        if x.ndim > 1:
            self.num_evaluations += x.shape[0]
            return self.loglike_rows(x)
        self.num_evaluations += 1
        return self.loglike_rows(x[None, :])[0]

    def loglike(self, x):
        return self.loglike_rows(np.asarray(x)[None, :])[0]

    def loglike_rows(self, x):
        raise NotImplementedError


class Rosenbrock(Likelihood):
    """likelihoods.py:48-59"""
    hip_like_id = LIKE_IDS['rosenbrock']

    def loglike_rows(self, x):
        # the reference adds the terms with Python's `sum`, i.e. left to right in the dtype of x (likelihoods.py:51); cumsum
        # is numpy's left-to-right accumulation (np.sum would add pairwise: other bits in float32 from 8 terms on)
        terms = 100.0 * (x[:, 1:] - x[:, :-1] ** 2.0) ** 2.0 + (1 - x[:, :-1]) ** 2.0
        return -np.cumsum(terms, axis=1)[:, -1]

    @property
    def max_loglike(self):
        return self(np.ones((self.x_dim,)))


class Himmelblau(Likelihood):
    """likelihoods.py:62-74 for x_dim = 2.  For even x_dim > 2 (BASELINE config 4 names x_dim=32, which the
    reference cannot run: it asserts x_dim == 2) the build-defined generalisation is the sum of the 2-D
    function over consecutive pairs (x[2i], x[2i+1]); it reduces to the reference at x_dim = 2."""
    hip_like_id = LIKE_IDS['himmelblau']

    def __init__(self, x_dim=2):
        assert x_dim >= 2 and x_dim % 2 == 0
        super(Himmelblau, self).__init__(x_dim)

    def loglike_rows(self, x):
        a, b = x[:, 0::2], x[:, 1::2]
        return np.sum(-(a ** 2 + b - 11.) ** 2 - (a + b ** 2 - 7.) ** 2, axis=1)

    @property
    def max_loglike(self):
        return self(np.array([3.0, 2.0] * (self.x_dim // 2)))


class GaussianMix(Likelihood):
    """likelihoods.py:165-193 with the defaults the fused kernel implements: sep=4, sigma=1, weights
    (0.4, 0.3, 0.2, 0.1)."""
    hip_like_id = LIKE_IDS['gaussmix']

    def __init__(self, x_dim, sep=4, weights=(0.4, 0.3, 0.2, 0.1), sigma=1):
        assert len(weights) in [2, 3, 4] and np.isclose(sum(weights), 1)
        super(GaussianMix, self).__init__(x_dim)
        self.sep, self.weights, self.sigma = sep, tuple(weights), sigma
        if not (sep == 4 and sigma == 1 and tuple(weights) == (0.4, 0.3, 0.2, 0.1)):
            self.hip_like_id = None  # only the defaults are in the kernel; other settings run on the host protocol
        pos = [(0, sep), (0, -sep), (sep, 0), (-sep, 0)]
        self.positions = [np.asarray(p) for p in pos[:len(weights)]]

    def loglike_rows(self, x):
        # operation for operation what the reference does per row (likelihoods.py:153-162, :182-189): shift the first two
        # coordinates, square, np.sum over the row in the dtype of x, then float64 constants and scipy's logsumexp
        x = np.asarray(x)
        ls = []
        for w, p in zip(self.weights, self.positions):
            d = np.array(x, copy=True)
            d[:, :2] -= p
            logl = -(np.sum(d ** 2, axis=1) / (2 * self.sigma ** 2))
            logl = logl - np.log(2 * np.pi * (self.sigma ** 2)) * self.x_dim / 2.0
            ls.append(logl + np.log(w))
        return scipy.special.logsumexp(np.stack(ls, axis=0), axis=0)

    @property
    def max_loglike(self):
        v = np.zeros(self.x_dim)
        v[:2] = self.positions[int(np.argmax(self.weights))]
        return self(v)


class Gaussian(Likelihood):
    """likelihoods.py:77-94: N(0, Sigma), Sigma = I + corr (11^T - I).  Evaluated in closed form (Sherman-Morrison
    for the equicorrelated covariance) instead of scipy's Cholesky; same float64 arithmetic domain."""
    hip_like_id = LIKE_IDS['gaussian']

    def __init__(self, x_dim, corr, lim=5):
        super(Gaussian, self).__init__(x_dim)
        self.corr, self.lim = corr, lim
        self.hip_like_params = (corr,)

    def loglike_rows(self, x):
        x = np.asarray(x, dtype=np.float64)
        D, c = self.x_dim, self.corr
        s1, s2 = np.sum(x, axis=1), np.sum(x * x, axis=1)
        quad = (s2 - c * s1 * s1 / (1.0 + (D - 1.0) * c)) / (1.0 - c)
        logdet = (D - 1.0) * np.log(1.0 - c) + np.log(1.0 + (D - 1.0) * c)
        return -0.5 * quad - 0.5 * logdet - 0.5 * D * np.log(2 * np.pi)

    @property
    def max_loglike(self):
        return self(np.zeros(self.x_dim))


class Eggbox(Likelihood):
    """likelihoods.py:97-110 (x_dim = 2)"""
    hip_like_id = LIKE_IDS['eggbox']

    def __init__(self, x_dim=2):
        assert x_dim == 2
        super(Eggbox, self).__init__(x_dim)

    def loglike_rows(self, x):
        chi = np.cos(x[:, 0] / 2.) * np.cos(x[:, 1] / 2.)
        return (2. + chi) ** 5

    @property
    def max_loglike(self):
        return self(np.zeros(2))


class GaussianShell(Likelihood):
    """likelihoods.py:113-132 (the fused kernel takes a scalar centre)"""
    hip_like_id = LIKE_IDS['shell']

    def __init__(self, x_dim, sigma=0.1, rshell=2, center=0):
        super(GaussianShell, self).__init__(x_dim)
        self.sigma, self.rshell = sigma, rshell
        if hasattr(center, '__len__'):
            self.center = np.asarray(center, dtype=np.float64)
            if np.ptp(self.center) != 0:
                self.hip_like_id = None
            c0 = float(self.center[0])
        else:
            self.center = np.full(x_dim, float(center))
            c0 = float(center)
        self.hip_like_params = (sigma, rshell, c0)

    def loglike_rows(self, x):
        rad = np.sqrt(np.sum((self.center - np.asarray(x, dtype=np.float64)) ** 2, axis=1))
        return -((rad - self.rshell) ** 2) / (2 * self.sigma ** 2)

    @property
    def max_loglike(self):
        return 0.0


class DoubleGaussianShell(Likelihood):
    """likelihoods.py:135-150"""
    hip_like_id = LIKE_IDS['double_shell']

    def __init__(self, x_dim, sigmas=(0.1, 0.1), rshells=(2, 2), centers=(-4, 4), weights=(1.0, 1.0)):
        super(DoubleGaussianShell, self).__init__(x_dim)
        self.shell1 = GaussianShell(x_dim, sigma=sigmas[0], rshell=rshells[0], center=centers[0])
        self.shell2 = GaussianShell(x_dim, sigma=sigmas[1], rshell=rshells[1], center=centers[1])
        self.weights = tuple(weights)
        if self.weights != (1.0, 1.0):
            self.hip_like_id = None
        self.hip_like_params = (sigmas[0], rshells[0], centers[0], sigmas[1], rshells[1], centers[1])

    def loglike_rows(self, x):
        return np.logaddexp(np.log(self.weights[0]) + self.shell1.loglike_rows(x),
                            np.log(self.weights[1]) + self.shell2.loglike_rows(x))
