"""CPU tests of the host driver (nnest_amd.sampler / nnest_amd.nested) with the TEST-ONLY oracle-backed
trainer: the reference's own integration check (reference tests/test_nested.py:10-19: Rosenbrock 2-D,
|logZ + 5.80| <= 0.2) and bookkeeping invariants of the evidence loop (nested.py:244-293, :458-506)."""
import csv
import os

import numpy as np
import pytest
import torch

from nnest_amd.nested import NestedSampler
from nnest_amd.sampler import detect_linear_scale
from nnest_amd.likelihoods import Rosenbrock, GaussianMix, Himmelblau
from nnest_amd.priors import UniformPrior
from tests.oracle_trainer import OracleTrainer


def test_detect_linear_scale():
    assert detect_linear_scale(lambda x: 5 * x, 7) == 5.0
    assert detect_linear_scale(None, 3) == 1.0
    assert detect_linear_scale(lambda x: x * 5 * np.pi, 2) == 5 * np.pi
    assert detect_linear_scale(lambda x: x ** 3, 2) is None
    assert detect_linear_scale(lambda x: x + 1, 2) is None


def test_uniform_prior_protocol():
    p = UniformPrior(3, -1, 1)
    assert p(np.array([0.0, 1.0, -1.0])) == 0
    assert p(np.array([0.0, 1.0000001, 0.0])) == -np.inf
    assert p(np.array([np.nan, 0.0, 0.0])) == 0  # NaN compares false, as in the reference (priors.py:39-43)
    np.random.seed(0)
    s = p.sample(5)
    np.random.seed(0)
    assert np.array_equal(s, -1 + 2 * np.random.uniform(size=(5, 3)))
    assert p.is_unit_box()


def test_likelihood_protocol_counts_and_shapes():
    like = Rosenbrock(4)
    x = np.random.RandomState(0).uniform(-5, 5, size=(7, 4))
    out = like(x)
    assert out.shape == (7,) and like.num_evaluations == 7
    assert np.isclose(like(x[0]), out[0]) and like.num_evaluations == 8
    assert like.max_loglike == 0
    assert np.isclose(Himmelblau(2).max_loglike, 0)
    assert np.isclose(Himmelblau(4)(np.array([3.0, 2.0, 3.0, 2.0])), 0)
    g = GaussianMix(3)
    assert g.hip_like_id is not None and GaussianMix(3, sep=5).hip_like_id is None


def test_nested_rosenbrock_2d_logz(tmp_path):
    """reference tests/test_nested.py: logZ = -5.80 +- 0.2 (there with 1000 live points; 400 here for speed,
    statistical error sqrt(h/N) ~ 0.11)."""
    np.random.seed(0)
    torch.manual_seed(0)
    tr = OracleTrainer(2, seed=0)
    s = NestedSampler(2, Rosenbrock(2), transform=lambda x: 5 * x, log_dir=str(tmp_path), num_live_points=400,
                      trainer=tr, log_level=30)
    s.run(train_iters=200, mcmc_num_chains=10, mcmc_dynamic_step_size=False)
    assert abs(s.logz + 5.80) <= 0.3, s.logz
    assert tr.num_trains >= 1 and s.num_batches > 0
    # bookkeeping: weights normalised, sample count = dead + live points, files written in the reference layout
    assert abs(np.sum(s.weights) - 1.0) < 1e-8
    assert s.samples.shape == (s.niter - 1 + 400, 2)
    run = s.logs['run_dir']
    with open(os.path.join(run, 'results', 'final.csv')) as f:
        rows = list(csv.reader(f))
    assert rows[0] == ['niter', 'ncall', 'logz', 'logzerr', 'h'] and float(rows[1][2]) == s.logz
    assert os.path.exists(os.path.join(run, 'chains', 'chain.txt'))
    assert os.path.exists(os.path.join(run, 'checkpoint', 'checkpoint_0.txt'))
    chain = np.loadtxt(os.path.join(run, 'chains', 'chain.txt'))
    assert chain.shape == (s.samples.shape[0], 2 + 2)
    # posterior mean of Rosenbrock 2-D on [-5,5]^2 is near (0.7..1.1, 1..1.7) (golden: 1.04, 1.51)
    mean = np.sum(s.samples * s.weights[:, None], 0)
    assert 0.3 < mean[0] < 1.6 and 0.6 < mean[1] < 2.4


def test_nested_resume_from_checkpoint(tmp_path):
    np.random.seed(1)
    torch.manual_seed(1)
    like = Rosenbrock(2)
    kw = dict(transform=lambda x: 5 * x, log_dir=str(tmp_path), num_live_points=100, log_level=30, append_run_num=False,
              checkpoint_min_seconds=0.0)
    s = NestedSampler(2, like, trainer=OracleTrainer(2, seed=1), **kw)
    s.run(train_iters=50, mcmc_num_chains=10, max_iters=150, strategy=['mcmc'])
    cps = sorted(int(f.split('_')[1].split('.')[0]) for f in os.listdir(os.path.join(str(tmp_path), 'checkpoint'))
                 if f.startswith('checkpoint_'))
    assert cps[-1] >= 140
    s2 = NestedSampler(2, like, trainer=OracleTrainer(2, seed=1), **kw)
    assert not s2.logs['created']
    s2.run(train_iters=50, mcmc_num_chains=10, strategy=['mcmc'])
    assert abs(s2.logz + 5.80) <= 3 * max(s2.logzerr, 0.25)  # 100 live points, 50 training epochs: sqrt(h/N) ~ 0.23, run-to-run scatter 0.33
    assert s2.niter > cps[-1]


def test_derived_parameters_follow_their_points(tmp_path):
    """loglike -> (logl, derived) (sampler.py:118-133): through prior rejection (nested.py:368-369), MCMC (nested.py:436-437) and
    into the saved chain (nested.py:287-288), on the host protocol with the oracle-backed trainer"""
    np.random.seed(4)
    torch.manual_seed(4)

    def like(x):
        logl = -(100.0 * (x[:, 1] - x[:, 0] ** 2) ** 2 + (1 - x[:, 0]) ** 2)
        return logl, np.stack([x[:, 0] - x[:, 1], 2.0 * x[:, 0]], axis=1)

    s = NestedSampler(2, like, transform=lambda x: 5 * x, log_dir=str(tmp_path), num_live_points=100, log_level=30, num_derived=2,
                      trainer=OracleTrainer(2, seed=4))
    s.run(train_iters=100, mcmc_num_chains=10)
    v = s.samples
    assert v.shape[1] == 4 and v.shape[0] == s.niter - 1 + 100
    np.testing.assert_allclose(v[:, 2], v[:, 0] - v[:, 1], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(v[:, 3], 2.0 * v[:, 0], rtol=1e-6, atol=1e-6)
    assert abs(s.logz + 5.80) <= 3 * max(s.logzerr, 0.25)
    with pytest.raises(ValueError):
        NestedSampler(2, like, transform=lambda x: 5 * x, log_dir=str(tmp_path / 'bad'), num_live_points=10, log_level=40,
                      num_derived=1, trainer=OracleTrainer(2, seed=4)).run(max_iters=5)
