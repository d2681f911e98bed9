"""Analytic test likelihoods with the reference's host protocol (nnest/likelihoods.py:7-22):
`like(x[N,D] or x[D]) -> logl`, `.x_dim`, `.num_evaluations`.  The host evaluation is vectorised numpy in
the dtype of the input (it is what the reference's Python row loop computes, used for the initial live
points, nested.py:228).  `hip_like_id` tells the sampler that the same function exists inside the fused
HIP kernels (include/nnest_hip.h NNEST_LIKE_*), which is where the MCMC evaluations happen."""
import numpy as np

from ._lib import LIKE_IDS


class Likelihood(object):
    num_derived = 0
    hip_like_id = None

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
        return -np.sum(100.0 * (x[:, 1:] - x[:, :-1] ** 2.0) ** 2.0 + (1 - x[:, :-1]) ** 2.0, axis=1)

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
        x = np.asarray(x)
        rest = np.sum(x[:, 2:] ** 2, axis=1)
        ls = []
        for w, p in zip(self.weights, self.positions):
            s = rest + (x[:, 0] - p[0]) ** 2 + (x[:, 1] - p[1]) ** 2
            ls.append(-(s / (2 * self.sigma ** 2)).astype(np.float64) - np.log(2 * np.pi * self.sigma ** 2) * self.x_dim / 2.0
                      + np.log(w))
        ls = np.stack(ls, axis=0)
        mx = np.max(ls, axis=0)
        return mx + np.log(np.sum(np.exp(ls - mx), axis=0))

    @property
    def max_loglike(self):
        v = np.zeros(self.x_dim)
        v[:2] = self.positions[int(np.argmax(self.weights))]
        return self(v)
