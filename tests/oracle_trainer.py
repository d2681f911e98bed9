"""TEST-ONLY Trainer-protocol object backed by the oracle (CPU).  It lets the host driver
(nnest_amd.sampler / nnest_amd.nested) run without a GPU in the `-m "not gpu"` suite and provides the
"CPU path" that GPU runs are compared against.  Never imported by the product package."""
import numpy as np
import torch

from oracle import oracle as orc
from nnest_amd.utils import ScalarWriter


class _Net(object):
    def __init__(self, nvp):
        self.nvp = nvp

    def eval(self):
        return self

    def store_packed(self):
        return self.nvp.w.copy()

    def load_packed(self, w):
        self.nvp.w[:] = w


class _Lazy(object):
    """per-epoch randomness drawn when the epoch is reached (early stopping leaves most of max_iters unused)"""

    def __init__(self, draw):
        self.draw, self.at, self.cur = draw, -1, None

    def __getitem__(self, i):
        assert i >= self.at
        while self.at < i:
            self.cur, self.at = self.draw(), self.at + 1
        return self.cur


class OracleTrainer(object):
    def __init__(self, x_dim, hidden_dim=16, num_blocks=3, num_layers=1, batch_size=100, learning_rate=1e-3,
                 weight_decay=1e-6, seed=0):
        rng = np.random.RandomState(seed)
        self.x_dim = x_dim
        self.nvp = orc.NVP(x_dim, hidden_dim, num_blocks, num_layers)
        # nn.Linear default init: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weights and biases
        w, D, H, L = [], x_dim, hidden_dim, num_layers
        for _ in range(num_blocks * 2):
            for (o, k) in [(H, D)] + [(H, H)] * L + [(D, H)]:
                b = 1.0 / np.sqrt(k)
                w.append(rng.uniform(-b, b, size=o * k))
                w.append(rng.uniform(-b, b, size=o))
        self.nvp.w[:] = np.concatenate(w).astype(np.float32)
        self.netG = _Net(self.nvp)
        self.device = torch.device('cpu')
        self.writer = ScalarWriter(None)
        self.batch_size, self.lr, self.wd = batch_size, learning_rate, weight_decay
        self.num_trains = 0

    def forward(self, x, to_numpy=False):
        x = x.numpy() if torch.is_tensor(x) else np.asarray(x)
        z, ld = self.nvp.forward(x.astype(np.float32))
        return (z, ld) if to_numpy else (torch.from_numpy(z), torch.from_numpy(ld))

    def inverse(self, z, to_numpy=False):
        z = z.numpy() if torch.is_tensor(z) else np.asarray(z)
        x, ld = self.nvp.inverse(z.astype(np.float32))
        return (x, ld) if to_numpy else (torch.from_numpy(x), torch.from_numpy(ld))

    def get_samples(self, z, to_numpy=False):
        return self.inverse(z, to_numpy=to_numpy)[0]

    def get_latent_samples(self, x, to_numpy=False):
        return self.forward(x, to_numpy=to_numpy)[0]

    def get_prior_samples(self, n, to_numpy=False):
        z = torch.randn(n, self.x_dim)
        return z.numpy() if to_numpy else z

    def train(self, samples, max_iters=10000, jitter=0.0, validation_fraction=0.1, patience=50, rng_seed=None, **kw):
        if rng_seed is not None:   # a replicated retrain (nnest_amd/nested.py::_train): every rank draws the same split / permutations / noise
            saved = np.random.get_state()
            np.random.seed(int(rng_seed) & 0x7FFFFFFF)
            try:
                return self.train(samples, max_iters=max_iters, jitter=jitter, validation_fraction=validation_fraction, patience=patience, **kw)
            finally:
                np.random.set_state(saved)
        samples = np.asarray(samples)
        N = samples.shape[0]
        if jitter < 0:
            jitter = orc.training_jitter(samples)
        n_valid = int(np.ceil(validation_fraction * N))
        n_train = N - n_valid
        split = np.random.permutation(N)
        perms = _Lazy(lambda: np.random.permutation(n_train).astype(np.int32))
        noises = _Lazy(lambda: np.random.normal(size=(n_train, self.x_dim)).astype(np.float32))
        res = self.nvp.train(samples, split, perms, noises, jitter, max_iters, patience=patience, batch=self.batch_size,
                             lr=self.lr, wd=self.wd, validation_fraction=validation_fraction)
        self.best_validation_loss = res['best_validation_loss']
        self.best_validation_epoch = res['best_validation_epoch']
        self.num_trains += 1
        self.total_iters = getattr(self, 'total_iters', 0) + res['epochs_run']
