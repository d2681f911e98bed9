"""Priors with the reference's protocol (nnest/priors.py): callable on ONE row -> log density; .sample(n)."""
import numpy as np


class Prior(object):

    def __init__(self, x_dim):
        self.x_dim = x_dim

    def __call__(self, x):
        raise NotImplementedError

    def sample(self, num_samples):
        raise NotImplementedError


class UniformPrior(Prior):
    """Box prior (nnest/priors.py:25-47): 0 inside [minimum, maximum], -inf outside; NaN counts as inside
    because both comparisons are false -- the HIP kernels reproduce exactly that (flow_tile.h inbox_tile)."""

    def __init__(self, x_dim, minimum, maximum):
        self.minimum = np.array([minimum] * x_dim) if not hasattr(minimum, '__len__') else np.array(minimum)
        self.maximum = np.array([maximum] * x_dim) if not hasattr(maximum, '__len__') else np.array(maximum)
        assert len(self.minimum) == x_dim and len(self.maximum) == x_dim
        super(UniformPrior, self).__init__(x_dim)

    def __call__(self, x):
        if np.any(x < self.minimum) or np.any(x > self.maximum):
            return -np.inf
        return 0

    def log_prob_rows(self, x):
        """__call__ over the rows of x[N, D] at once (same comparisons, so NaN rows count as inside too)"""
        out_of_box = np.any(x < self.minimum, axis=1) | np.any(x > self.maximum, axis=1)
        return np.where(out_of_box, -np.inf, 0.0)

    def sample(self, num_samples):
        # numpy global RNG, one uniform block of (n, D) -- same stream consumption as priors.py:45-47
        return self.minimum + (self.maximum - self.minimum) * np.random.uniform(size=(num_samples, self.x_dim))

    def is_unit_box(self):
        return bool(np.all(self.minimum == -1) and np.all(self.maximum == 1))
