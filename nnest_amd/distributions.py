"""GeneralisedNormal: the reference's optional base distribution of the flow (nnest/distributions/generalised_normal.py;
examples/nested/run.py:20-21 `--base_dist gen_normal --beta 8`): density  beta / (2 scale Gamma(1/beta)) exp(-(|x - loc| / scale)^beta)
per dimension.  With beta = 8 it is close to uniform on [-1, 1], which is what the 'rejection_flow' strategy's `usample`
relies on (nnest/sampler.py:575-576).

The kernels implement loc = 0, scale = 1 (the only values the reference's front-ends construct); `Trainer(base_dist=...)`
accepts this class or the reference's own (duck-typed on `beta`, `loc`, `scale`, `usample`)."""
import math

import numpy as np
import torch


class GeneralisedNormal(object):

    def __init__(self, loc, scale, beta):
        self.loc = torch.as_tensor(loc, dtype=torch.float32)
        self.scale = torch.as_tensor(scale, dtype=torch.float32)
        self.beta = beta

    @property
    def mean(self):
        return self.loc

    def _beta(self):
        return float(self.beta.item()) if torch.is_tensor(self.beta) else float(self.beta)

    def sample(self, sample_shape=torch.Size()):
        """generalised_normal.py:50-53: scipy.stats.gennorm.rvs(beta) (loc / scale are not applied there either)"""
        from scipy.stats import gennorm
        shape = tuple(sample_shape) + tuple(self.loc.shape)
        return torch.tensor(gennorm.rvs(self._beta(), size=shape), dtype=torch.float32)

    def usample(self, sample_shape=torch.Size()):
        """generalised_normal.py:58-60: uniform on [-1, 1]^D (numpy global RNG)"""
        shape = tuple(sample_shape) + tuple(self.loc.shape)
        return 2 * (np.random.uniform(size=shape) - 0.5)

    def log_prob(self, value):
        """generalised_normal.py:62-68 (per dimension)"""
        b = self._beta()
        return (-((torch.abs(value - self.loc) / self.scale) ** b) + math.log(b) - torch.log(self.scale) - math.log(2)
                - math.lgamma(1.0 / b))
