"""GPU end-to-end tests: NestedSampler on the HIP path (fused persistent-MH kernel + on-device training).

log Z checks.  Rosenbrock 2-D on [-5,5]^2 has the closed form  Z = (pi/10) * (1 - erfc((sqrt(5)-1))/2) / 100,
log Z = -5.804; the reference's own integration test accepts |logZ + 5.80| <= 0.2 (reference
tests/test_nested.py:7, :18-19).  BASELINE.json asks for |logZ_GPU - logZ_CPU| <= 0.1 on the same seeds:
chains driven by different random streams are independent estimates with standard error sqrt(h/N) each, so
that comparison is made on means over several seeds, with the tolerance stated below.
"""
import json
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

from nnest_amd.likelihoods import Rosenbrock, GaussianMix, Himmelblau  # noqa: E402
from nnest_amd.nested import NestedSampler  # noqa: E402

LOGZ_ROSEN2D = math.log(math.pi / 10 * (1 - 0.5 * math.erfc(math.sqrt(5) - 1)) / 100)
G = os.path.join(os.path.dirname(__file__), 'golden')


def run(tmp, D, like, scale, N, seed, **kw):
    np.random.seed(seed)
    torch.manual_seed(seed)
    s = NestedSampler(D, like, transform=lambda x: scale * x, log_dir=str(tmp), num_live_points=N, log_level=30,
                      hidden_dim=16, num_blocks=3, num_layers=1, flow=kw.pop('flow', 'nvp'))
    assert s._fused_like_id is not None, 'fused HIP path not selected'
    s.run(**kw)
    return s


def test_closed_form():
    assert abs(LOGZ_ROSEN2D + 5.804) < 1e-3


def test_config1_rosenbrock_2d_100_live_points(tmp_path):
    """BASELINE config 1 (the reference's CPU-runnable case) with the reference's arguments; reference result
    on seeds 0/0: logZ = -6.026 +- 0.227 (tests/golden/nested_cfg1.json)."""
    with open(os.path.join(G, 'nested_cfg1.json')) as f:
        ref = json.load(f)
    logzs = []
    for seed in range(6):
        s = run(tmp_path / ('s%d' % seed), 2, Rosenbrock(2), 5.0, 100, seed, train_iters=2000, mcmc_num_chains=10)
        logzs.append(s.logz)
        assert abs(s.logzerr - ref['logzerr']) < 0.08
    m, se = float(np.mean(logzs)), float(np.std(logzs) / np.sqrt(len(logzs)))
    # mean of 6 runs: standard error ~0.23/sqrt(6) = 0.09; within 0.25 of the closed form and of the reference run
    assert abs(m - LOGZ_ROSEN2D) < 0.25, (m, se, logzs)
    assert abs(m - ref['logz']) < 0.45, (m, ref['logz'])


def test_rosenbrock_2d_1000_live_points_reference_test(tmp_path):
    """reference tests/test_nested.py:10-19 (with flow='nvp'): 1000 live points, 10 chains, fixed step."""
    s = run(tmp_path, 2, Rosenbrock(2), 5.0, 1000, 0, mcmc_num_chains=10, mcmc_dynamic_step_size=False)
    assert abs(s.logz - LOGZ_ROSEN2D) <= 0.2, s.logz
    assert abs(np.sum(s.weights) - 1) < 1e-8
    mean = np.sum(s.samples * s.weights[:, None], 0)
    assert abs(mean[0] - 1.0) < 0.15 and abs(mean[1] - 1.5) < 0.25  # truncated-box posterior mean ~ (0.96, 1.43)


def test_wide_batch_matches_narrow_batch(tmp_path):
    """One walker per live point (the bench configuration) vs the reference's 10 chains: same evidence."""
    a = run(tmp_path / 'wide', 2, Rosenbrock(2), 5.0, 1000, 1, mcmc_num_chains=1000)
    assert abs(a.logz - LOGZ_ROSEN2D) <= 0.2, a.logz
    assert a.num_batches < 40


def test_gaussian_mixture_and_himmelblau(tmp_path):
    # 2-D mixture of 4 unit Gaussians inside [-10,10]^2: Z = 1/400 up to tails -> logZ = -5.991
    s = run(tmp_path / 'gm', 2, GaussianMix(2), 10.0, 500, 2, mcmc_num_chains=50)
    assert abs(s.logz - math.log(1 / 400.0)) <= 0.25, s.logz
    # Himmelblau on [-5,5]^2: four modes; numerical integral of exp(-f)/100
    xs = np.linspace(-5, 5, 2001)
    X, Y = np.meshgrid(xs, xs)
    Z = np.sum(np.exp(-(X ** 2 + Y - 11) ** 2 - (X + Y ** 2 - 7) ** 2)) * (xs[1] - xs[0]) ** 2 / 100
    s = run(tmp_path / 'hb', 2, Himmelblau(2), 5.0, 500, 3, mcmc_num_chains=50)
    assert abs(s.logz - math.log(Z)) <= 0.3, (s.logz, math.log(Z))


def test_config2_slice_rosenbrock_50d(tmp_path):
    """BASELINE config 2 (Rosenbrock x_dim=50, 1000 live points): a bounded slice of the run on the fused path;
    checks the run mechanics (retrain cadence nested.py:311-314, batch consumption nested.py:429-439)."""
    s = run(tmp_path, 50, Rosenbrock(50), 5.0, 1000, 0, strategy=['mcmc'], mcmc_num_chains=1000, max_iters=1500,
            train_iters=60)
    assert s.niter >= 1500
    assert s.num_retrains == 1 + 1500 // 500
    assert 2 <= s.num_batches <= 12
    assert np.isfinite(s.logz) and s.ncall > 1000
    assert np.all(np.diff(s.loglikes[:1500]) >= 0)  # dead points leave in increasing likelihood order


def test_user_callable_likelihood_runs_on_the_host_protocol(tmp_path):
    """A plain Python `loglike` (the reference's plugin protocol, sampler.py:110-133) is not in the kernels: the flow
    passes run on the GPU, the likelihood and the accept logic on the host (Sampler._mcmc_sample_host)."""
    np.random.seed(3)
    torch.manual_seed(3)

    def rosen(x):  # x: (N, 2) already transformed
        return -(100.0 * (x[:, 1] - x[:, 0] ** 2) ** 2 + (1 - x[:, 0]) ** 2)

    s = NestedSampler(2, rosen, transform=lambda x: 5 * x, log_dir=str(tmp_path), num_live_points=300, log_level=30, flow='nvp')
    assert s._fused_like_id is None
    s.run(mcmc_num_chains=20, train_iters=200)
    assert abs(s.logz - LOGZ_ROSEN2D) <= 0.35, s.logz   # 300 live points: sqrt(h/N) ~ 0.13
    assert s.total_calls > 300


def test_trainer_chunked_launches_equal_one_launch(tmp_path):
    """Trainer.train issues a long run as 128-epoch launches that carry the early-stopping state; with the same
    split / shuffles / noise seed the result is bit-identical to a single launch (include/nnest_hip.h)."""
    from nnest_amd.trainer import Trainer
    from nnest_amd import trainer as trainer_mod
    rng = np.random.RandomState(0)
    D, N, E = 6, 220, 300
    live = rng.normal(size=(N, D)) * 0.3
    split = rng.permutation(N)
    perms = np.stack([rng.permutation(N - 22) for _ in range(E)]).astype(np.int32)
    outs = []
    for chunk in (128, 1000):
        trainer_mod.EPOCH_CHUNK = chunk
        torch.manual_seed(7)
        t = Trainer(D, log_dir=None, learning_rate=1e-3, seed=1, log_level=30, flow='nvp')
        t.train(live, max_iters=E, jitter=0.01, split=split, perms=perms, patience=40)
        outs.append((t.netG.store_packed(), t.best_validation_epoch, t.best_validation_loss, t.total_iters, t.losses.copy()))
    trainer_mod.EPOCH_CHUNK = 128
    assert np.array_equal(outs[0][0], outs[1][0])
    assert outs[0][1:4] == outs[1][1:4]
    assert np.array_equal(outs[0][4], outs[1][4])
    assert outs[0][3] <= E and outs[0][1] >= 1
    # jitter < 0: the k-d tree rule of the reference (trainer.py:168-171), computed on the device
    from oracle import oracle as orc
    assert abs(t.training_jitter(live) - orc.training_jitter(live)) < 1e-12


@pytest.mark.parametrize('strategy', [['rejection_prior', 'rejection_flow', 'mcmc'], ['rejection_prior', 'density_flow', 'mcmc'],
                                      ['rejection_flow']])
def test_flow_rejection_strategies(tmp_path, strategy):
    """'rejection_flow' (sampler.py:545-605) and 'density_flow' (sampler.py:607-628): blocks of candidates per launch;
    the evidence must come out the same as with the default strategy pair."""
    s = run(tmp_path, 2, Rosenbrock(2), 5.0, 400, 11, strategy=strategy, mcmc_num_chains=40, train_iters=300)
    # 'density_flow' draws replacements from the flow's density, not uniformly inside the likelihood contour
    # (sampler.py:607-628 as written): its evidence is biased by construction, so only a loose bound applies to it
    tol = 1.0 if 'density_flow' in strategy else 0.35        # sqrt(h/N) ~ 0.11
    assert abs(s.logz - LOGZ_ROSEN2D) <= tol, s.logz
    assert s.num_retrains >= 1


def test_flow_rejection_with_host_likelihood(tmp_path):
    np.random.seed(5)
    torch.manual_seed(5)
    s = NestedSampler(2, lambda x: -(100.0 * (x[:, 1] - x[:, 0] ** 2) ** 2 + (1 - x[:, 0]) ** 2), transform=lambda x: 5 * x,
                      log_dir=str(tmp_path), num_live_points=200, log_level=30, flow='nvp')
    assert s._fused_like_id is None
    s.run(strategy=['rejection_prior', 'rejection_flow'], train_iters=200, max_iters=900)
    x, logl, derived, ncall = s._density_sample(float(np.median(s.loglikes[-200:])))
    assert x.shape == (1, 2) and logl.shape == (1,) and ncall >= 1 and logl[0] > np.median(s.loglikes[-200:])
    assert np.isfinite(s.logz)


def test_spline_flow_rosenbrock_2d_reference_integration_test(tmp_path):
    """The reference's own integration test (tests/test_nested.py:10-19) on its default flow, the neural spline flow:
    Rosenbrock 2-D, 1000 live points, 10 chains, fixed step; |logZ + 5.80| <= 0.2."""
    s = run(tmp_path, 2, Rosenbrock(2), 5.0, 1000, 0, flow='spline', mcmc_num_chains=10, mcmc_dynamic_step_size=False)
    from nnest_amd.spline import HipSpline
    assert isinstance(s.trainer.netG, HipSpline) and s.trainer.netG.data_dep_init_done
    assert abs(s.logz - LOGZ_ROSEN2D) <= 0.2, s.logz
    mean = np.sum(s.samples * s.weights[:, None], 0)
    assert abs(mean[0] - 1.0) < 0.15 and abs(mean[1] - 1.5) < 0.25


def test_spline_flow_default_and_wide_batch(tmp_path):
    """flow='spline' is the default of the reference (nested.py:35) and of this build; one walker per live point."""
    np.random.seed(4)
    torch.manual_seed(4)
    s = NestedSampler(2, Rosenbrock(2), transform=lambda x: 5 * x, log_dir=str(tmp_path), num_live_points=500, log_level=30)
    assert type(s.trainer.netG).__name__ == 'HipSpline' and s._fused_like_id is not None
    s.run(mcmc_num_chains=500, train_iters=300)
    assert abs(s.logz - LOGZ_ROSEN2D) <= 0.3, s.logz


def test_spline_flow_gaussian_mixture_5d(tmp_path):
    # 5-D mixture of 4 unit Gaussians inside [-10,10]^5: Z = 20^-5 up to tails
    s = run(tmp_path, 5, GaussianMix(5), 10.0, 500, 6, flow='spline', mcmc_num_chains=100, train_iters=300)
    assert abs(s.logz - 5 * math.log(1 / 20.0)) <= 0.5, s.logz


def test_spline_flow_reproduces_published_rosenbrock_10d_evidence(tmp_path):
    """The reference's published run (examples/nested/example.ipynb:869-890): Rosenbrock 10-D, 1000 live points,
    spline flow (3 blocks, hidden 16), rejection_prior -> mcmc, 50 MCMC steps: logZ = -43.364 +- 0.193, H = 37.2.
    Three seeds under the reference's step rule (batch-wide, lag 0): the mean has standard error 0.19 / sqrt(3), the
    published value 0.19 of its own; the bound is 2.5 sigma of the difference.  (With 50 steps a chain is not fully
    decorrelated, so the evidence moves with the step rule by ~0.1-0.3: tools/step_rule_study.py.)"""
    zs = []
    for seed in range(3):
        s = run(tmp_path / str(seed), 10, Rosenbrock(10), 5.0, 1000, seed, flow='spline', mcmc_steps=50, mcmc_num_chains=100,
                mcmc_step_lag=0)
        assert abs(s.logzerr - 0.193) < 0.03
        assert abs(s.h - 37.226) < 4.0
        zs.append(s.logz)
    assert abs(np.mean(zs) - (-43.364)) <= 2.5 * 0.193 * math.sqrt(1 + 1 / 3.0), zs


def test_mcmc_sampler_front_end(tmp_path):
    """MCMCSampler.run (mcmc.py:79-130): train on samples, then unconstrained Metropolis in the latent space; a 2-D
    correlated Gaussian target must come back with its mean and covariance"""
    import nnest_amd
    rng = np.random.RandomState(0)
    cov = np.array([[1.0, 0.8], [0.8, 1.0]])
    icov = np.linalg.inv(cov)
    train = rng.multivariate_normal([1.0, -2.0], cov, size=2000)

    def loglike(x):
        d = x - np.array([1.0, -2.0])
        return -0.5 * np.einsum('ni,ij,nj->n', d, icov, d)

    np.random.seed(1)
    torch.manual_seed(1)
    s = nnest_amd.MCMCSampler(2, loglike, log_dir=str(tmp_path), log_level=30, flow='nvp')
    s.run(400, 50, train, init_samples=(train[:50] - train.mean(0)) / train.std(0))
    assert s.samples.shape == (50, 401, 2) and s.loglikes.shape == (50, 401)
    flat = s.samples[:, 100:, :].reshape(-1, 2)
    assert np.all(np.abs(flat.mean(0) - np.array([1.0, -2.0])) < 0.25)
    c = np.cov(flat.T)
    assert abs(c[0, 0] - 1) < 0.3 and abs(c[1, 1] - 1) < 0.3 and abs(c[0, 1] - 0.8) < 0.3
    assert s.total_calls == 50 * 400 + 50   # the initial likelihoods of the chains count too (sampler.py:268-270)


def test_config4_himmelblau32_fused_likelihood_vs_oracle_and_slice(tmp_path):
    """BASELINE config 4 names Himmelblau at x_dim = 32; the reference asserts x_dim == 2 (likelihoods.py:62-67), so the
    form used here -- the 2-D function summed over consecutive pairs -- is build-defined: [UNPINNED beyond D = 2].  The
    fused kernels (K6, K3, K4) are checked against the oracle's restatement of that same sum, then a bounded slice of the
    nested run (4000 live points, one walker per live point) exercises the run mechanics."""
    from nnest_amd import flow
    from oracle import oracle as orc
    D = 32
    rng = np.random.RandomState(4)
    x = rng.uniform(-1, 1, size=(500, D)).astype(np.float32)
    lo = orc.loglike('himmelblau', x, 5.0)
    np.testing.assert_allclose(flow.loglike(2, x, 5.0).cpu().numpy(), lo, rtol=2e-6, atol=1e-4)           # K6
    nvp = flow.HipNVP(D, 16, 3, 1, seed=4)
    o = orc.NVP(D, 16, 3, 1, nvp.store_packed())
    z = (rng.normal(size=(300, D)) * 0.7).astype(np.float32)
    xg, ld, logl, inbox = nvp.inverse_loglike(2, 5.0, z)                                                   # K3
    np.testing.assert_allclose(logl.cpu().numpy(), orc.loglike('himmelblau', xg.cpu().numpy(), 5.0), rtol=2e-6, atol=1e-4)
    assert np.max(np.abs(xg.cpu().numpy() - o.inverse(z)[0])) < 5e-5
    for form in ('quad', 'team'):                                                                          # K4, both tile shapes
        init = rng.uniform(-0.8, 0.8, size=(64, D))
        il = orc.loglike('himmelblau', init, 5.0)
        zz, _ = nvp.forward(init)
        ll = torch.from_numpy(il).cuda()
        res = nvp.mh_steps(2, 5.0, zz, ll, float(il.min()) - 50.0, 0.05, 20, seed=3, history=True, form=form)
        hx = res['hist_x'].cpu().numpy()
        hl = res['hist_logl'].cpu().numpy()
        lo = orc.loglike('himmelblau', hx.reshape(-1, D), 5.0).reshape(hl.shape)
        assert np.max(np.abs(hl[:, 1:] - lo[:, 1:]) / (1 + np.abs(lo[:, 1:]))) < 5e-6
        assert np.all(hl[:, 1:] > float(il.min()) - 50.0) and int(res['n_accept'].sum()) > 0
    s = run(tmp_path, D, Himmelblau(D), 5.0, 4000, 0, strategy=['mcmc'], mcmc_num_chains=4000, max_iters=3000, train_iters=40)
    assert s.niter >= 3000 and s.num_retrains == 1 + 3000 // 2000 and np.isfinite(s.logz)
    assert np.all(np.diff(s.loglikes[:3000]) >= 0)


def test_config5_slice_rosenbrock_100d_8000_live_points(tmp_path):
    """BASELINE config 5 with the RealNVP flow (its MAF variant does not exist in the reference): x_dim 100, 8000 live points,
    one walker per live point -- a bounded slice of the run."""
    s = run(tmp_path, 100, Rosenbrock(100), 5.0, 8000, 0, strategy=['mcmc'], mcmc_num_chains=8000, max_iters=5000,
            train_iters=30)
    assert s.niter >= 5000 and s.num_retrains == 1 + 5000 // 4000
    assert np.isfinite(s.logz) and np.all(np.diff(s.loglikes[:5000]) >= 0)


def test_derived_parameters_host_protocol(tmp_path):
    """A likelihood that returns (logl, derived) (sampler.py:118-133): the derived columns follow their points through the
    prior-rejection phase, the MCMC phase and into the chain (nested.py:287-288, :368-369, :436-437)."""
    np.random.seed(11)
    torch.manual_seed(11)

    def like(x):
        logl = -(100.0 * (x[:, 1] - x[:, 0] ** 2) ** 2 + (1 - x[:, 0]) ** 2)
        return logl, np.stack([x[:, 0] + x[:, 1], x[:, 0] * x[:, 1]], axis=1)

    s = NestedSampler(2, like, transform=lambda x: 5 * x, log_dir=str(tmp_path), num_live_points=200, log_level=30, flow='nvp',
                      num_derived=2)
    assert s._fused_like_id is None
    s.run(mcmc_num_chains=20, train_iters=100)
    assert s.samples.shape[1] == 4
    v = s.samples
    np.testing.assert_allclose(v[:, 2], v[:, 0] + v[:, 1], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(v[:, 3], v[:, 0] * v[:, 1], rtol=1e-5, atol=1e-5)
    assert abs(s.logz - LOGZ_ROSEN2D) <= 0.5
    chain = np.loadtxt(os.path.join(s.logs['chains'], 'chain.txt'))
    assert chain.shape[1] == 2 + 4


def test_host_tensors_drop_in_level_1(tmp_path, monkeypatch):
    """INTEGRATION.md level 1: Trainer(host_tensors=True) under a sampler that mixes CPU tensors into the loop (the
    reference's _mcmc_sample, sampler.py:264-336, as restated in nnest_amd.sampler._mcmc_sample_host).  `device` reads cpu,
    every tensor handed out is a CPU tensor, and on the same recorded proposal noise the chain equals the device-tensor
    path bit for bit (the arithmetic is the same kernels)."""
    from nnest_amd.trainer import Trainer
    from nnest_amd.sampler import Sampler
    from nnest_amd.priors import UniformPrior
    rng = np.random.RandomState(2)
    init = rng.uniform(-0.5, 0.5, size=(24, 5))

    def like(x):
        return -np.sum(100.0 * (x[:, 1:] - x[:, :-1] ** 2.0) ** 2.0 + (1 - x[:, :-1]) ** 2.0, axis=1)

    out = {}
    noise = torch.randn(30, 24, 5, generator=torch.Generator().manual_seed(1))
    for host in (True, False):
        draws = list(noise)
        monkeypatch.setattr(torch, 'randn_like', lambda t: draws.pop(0).to(t.device))   # torch.randn_like(z), sampler.py:310
        tr = Trainer(5, flow='nvp', log_dir=str(tmp_path / ('h%d' % host)), host_tensors=host, seed=3, log_level=40)
        assert tr.device.type == ('cpu' if host else 'cuda')
        z, ld = tr.forward(init)
        x, ldi = tr.inverse(z)
        assert z.device.type == tr.device.type and x.device.type == tr.device.type
        assert (ldi.exp().clamp(max=1)).device.type == tr.device.type
        idx = (np.array([0, 2]),)
        assert x[idx].shape == (2, 5)                      # numpy index tuples, as the reference indexes (sampler.py:330)
        assert tr.get_prior_samples(3).device.type == tr.device.type
        s = Sampler(5, like, transform=lambda x: 5 * x, prior=UniformPrior(5, -1, 1), transform_prior=False, trainer=tr,
                    log_dir=str(tmp_path / ('s%d' % host)), log_level=40, fused=False)
        torch.manual_seed(7)
        np.random.seed(7)
        l0, _ = s.loglike(init)
        out[host] = s._mcmc_sample(30, step_size=0.05, dynamic_step_size=True, init_samples=init, init_loglikes=l0,
                                   init_derived=np.empty((24, 0)), loglstar=float(l0.min()) - 100.0)
    for a, b in zip(out[True][:4], out[False][:4]):
        assert np.array_equal(a, b)
    assert out[True][4] == out[False][4] and out[True][5] == out[False][5] and out[True][5] > 0


def _cpu_fixture(cfg):
    with open(os.path.join(G, 'logz_cpu_cfg%d.json' % cfg)) as f:
        return json.load(f)


def _fixture(kind, cfg):
    with open(os.path.join(G, 'logz_%s_cfg%d.json' % (kind, cfg))) as f:
        return json.load(f)


@pytest.mark.parametrize('cfg', [1, 2, 3, 11])
def test_committed_logz_fixtures_resolve_the_acceptance(cfg):
    """(cfg 11 = configuration 1 at the reference's default mcmc_num_chains = 10, nested.py:185, instead of one chain per live point.)
    BASELINE's acceptance "log Z within +-0.1 of the CPU reference" on the committed ensembles (CPU path:
    oracle/run_logz_cpu.py; GPU path: tools/run_logz_gpu.py; unpaired means over every seed each fixture holds): the means
    agree within 0.1 AND the comparison has the resolution to say so -- combined standard error <= 0.07, i.e. a true
    difference of 0.2 would stand out by three standard errors."""
    c, g = np.array(_fixture('cpu', cfg)['logz']), np.array(_fixture('gpu', cfg)['logz'])
    se = float(np.hypot(c.std(ddof=1) / np.sqrt(len(c)), g.std(ddof=1) / np.sqrt(len(g))))
    delta = float(g.mean() - c.mean())
    print('config %d: cpu %.3f (%d seeds), gpu %.3f (%d seeds), delta %.3f +- %.3f' % (cfg, c.mean(), len(c), g.mean(), len(g), delta, se))
    assert se <= 0.07, se
    assert abs(delta) <= 0.1, delta
    # the GPU fixture says which kernels produced it; the defaults a run of this configuration takes now must be those
    gfix = _fixture('gpu', cfg)
    b = gfix.get('build')
    assert b is not None, 'tests/golden/logz_gpu_cfg%d.json carries no build stamp: regenerate it (tools/run_logz_gpu.py)' % cfg
    from nnest_amd import flow as hipflow, _lib
    nvp = hipflow.HipNVP(gfix['x_dim'], 16, 3, 1, seed=0)
    C, steps = gfix['mcmc_num_chains'], 5 * gfix['x_dim']
    lag = nvp.default_lag(C) if steps >= 100 else 0
    warm = nvp.default_warm(C, 'batch', lag) if lag else 0
    assert b['abi_version'] == int(_lib.load().nnest_hip_version())
    assert (b['mh_form'], b['step_lag'], b['step_warm']) == (nvp.mh_form_for(C, dynamic='batch', lag=lag, warm=warm), lag, warm), b
    assert b['train_form'] == 'rows'


@pytest.mark.parametrize('cfg', [1, 2, 3, 11])
def test_logz_gpu_vs_cpu(tmp_path, cfg):
    """The GPU path re-run live (twelve seeds) against the FULL CPU-path fixture (the host driver on the oracle-backed trainer,
    oracle/run_logz_cpu.py -> tests/golden/logz_cpu_cfg<cfg>.json; same configuration and run() arguments).  The two paths
    draw from different noise streams, so a run is an independent estimate with scatter ~ sqrt(H/N) (0.2 at config 1, 0.45 at
    config 2, 0.12 at config 3).  Twelve live runs cannot resolve +-0.1 at config 2 by themselves (their mean scatters by
    0.18) -- that is the committed fixtures' job, test above; this test holds the live code to the fixtures: the live mean
    within three standard errors of its own sample from the CPU ensemble's mean, and from the committed GPU ensemble's."""
    ref, gfix = _fixture('cpu', cfg), _fixture('gpu', cfg)
    like = {'Rosenbrock': Rosenbrock, 'GaussianMix': GaussianMix}[ref['likelihood']](ref['x_dim'])
    scale = {'Rosenbrock': 5.0, 'GaussianMix': 10.0}[ref['likelihood']]
    seeds = ref['seeds'][:12]
    cpu, gcommitted = np.array(ref['logz']), np.array(gfix['logz'])
    gpu = []
    for seed in seeds:
        s = run(tmp_path / str(seed), ref['x_dim'], like, scale, ref['num_live_points'], seed, mcmc_num_chains=ref['mcmc_num_chains'])
        gpu.append(s.logz)
    gpu = np.array(gpu)
    se_live = float(gpu.std(ddof=1) / np.sqrt(len(gpu)))
    se = float(np.hypot(se_live, cpu.std(ddof=1) / np.sqrt(len(cpu))))
    delta = float(gpu.mean() - cpu.mean())
    print('config %d: live gpu %.3f +- %.3f (%d seeds), cpu fixture %.3f (%d seeds), delta %.3f +- %.3f; committed gpu %.3f' % (
        cfg, gpu.mean(), se_live, len(seeds), cpu.mean(), len(cpu), delta, se, gcommitted.mean()))
    assert abs(delta) <= 3 * se, (delta, se)
    if cfg == 3:   # a tight live check where one is cheap: twelve runs of config 3 resolve 0.04
        assert abs(delta) <= 0.1, (delta, se)
    assert abs(float(gpu.mean() - gcommitted.mean())) <= 3 * float(np.hypot(se_live, gcommitted.std(ddof=1) / np.sqrt(len(gcommitted))))
    # and the run-to-run scatter is the same on both sides (within a factor: 12 samples)
    assert 0.4 < gpu.std(ddof=1) / cpu.std(ddof=1) < 2.5


def test_native_prior_rejection_phase_equals_the_python_loop(tmp_path):
    """Round-5 verdict item 9: the 'rejection_prior' phase of run() (nnest/nested.py:322-334, :362-373 over
    Sampler._rejection_prior_sample, nnest/sampler.py:529-543) in the native library (nnest_host_prior_consume) beside the MCMC
    phase's loop.  The same seeded run through the native loop and through the Python loop: the kernels are deterministic, so the
    two must agree in EVERYTHING the run produces -- iterations, likelihood calls, log Z, H, every dead point -- with a volume
    switch, with the default efficiency switch, and when the run ends inside the prior phase."""
    for k, (kw, N) in enumerate([(dict(volume_switch=0.25, mcmc_num_chains=50), 300), (dict(mcmc_num_chains=50), 300),
                                 (dict(mcmc_num_chains=20, max_iters=150), 200), (dict(mcmc_num_chains=20, log_interval=7), 120)]):
        out = []
        for native in (True, False):
            np.random.seed(11 + k)
            torch.manual_seed(11 + k)
            s = NestedSampler(2, Rosenbrock(2), transform=lambda x: 5.0 * x, log_dir=str(tmp_path / ('%d_%s' % (k, native))),
                              num_live_points=N, log_level=30, flow='nvp', native_loop=native)
            assert s._fused_like_id is not None
            s.run(train_iters=100, **kw)
            out.append(s)
        a, b = out
        assert a.niter == b.niter and a.ncall == b.ncall and a.num_retrains == b.num_retrains and a.num_batches == b.num_batches
        assert a.logz == b.logz and a.h == b.h and a.logzerr == b.logzerr
        assert np.array_equal(a.samples, b.samples) and np.array_equal(a.loglikes, b.loglikes) and np.array_equal(a.weights, b.weights)
