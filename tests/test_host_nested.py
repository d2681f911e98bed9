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


@pytest.mark.parametrize('nd,resume_at', [(0, None), (1, None), (0, 'checkpoint')])
def test_native_loop_equals_the_python_loop(tmp_path, nd, resume_at):
    """The per-iteration body of run() in the native library (nnest_host_mcmc_consume: nested.py:269-293, :429-437, :458-471)
    against its Python restatement in nnest_amd/nested.py on the same seeds: log Z, H, iteration and call counts, every dead
    point, likelihood and weight, results.csv and the dead-point files of the checkpoints -- EQUAL, not close; with a derived
    parameter travelling with the points, and through a run that is stopped and resumed from its checkpoint under the native
    loop (the saved_*.npy files then start from the rows read back, GrowingNpy(initial=...))."""
    D = 3

    def like(x):
        x = np.atleast_2d(x)
        l = -0.5 * np.sum((x / 0.4) ** 2, axis=1)
        return (l, x[:, :1] ** 2) if nd else l

    def go(native, sub, **kw):
        np.random.seed(7)
        torch.manual_seed(7)
        tr = OracleTrainer(D, seed=3)
        s = NestedSampler(D, like, transform=lambda x: 3 * x, log_dir=str(tmp_path / sub), num_live_points=120, trainer=tr, log_level=30,
                          num_derived=nd, append_run_num=False, checkpoint_min_seconds=0.0, chain_min_seconds=0.0, native_loop=native,
                          fused=False)
        s.run(train_iters=20, mcmc_num_chains=8, mcmc_steps=10, log_interval=30, **kw)
        return s

    if resume_at is None:
        a, b = go(False, 'py'), go(True, 'native')
    else:
        a = go(False, 'py')
        go(True, 'native', max_iters=400)          # stops at the iteration cap, checkpoints on disk
        b = go(True, 'native')                      # resume=True: picks up the last checkpoint
        assert b.niter == a.niter or b.niter > 401  # (a resumed run draws fresh randomness: compared by its invariants below)
    if resume_at is None:
        assert a.logz == b.logz and a.h == b.h and a.niter == b.niter and a.ncall == b.ncall
        for k in ('samples', 'weights', 'loglikes'):
            assert np.array_equal(getattr(a, k), getattr(b, k)), k
        for name in ('results/results.csv', 'results/final.csv'):
            assert open(str(tmp_path / 'py' / name)).read() == open(str(tmp_path / 'native' / name)).read(), name
        cps = sorted(f for f in os.listdir(str(tmp_path / 'py' / 'checkpoint')) if f.startswith('checkpoint_'))
        assert cps == sorted(f for f in os.listdir(str(tmp_path / 'native' / 'checkpoint')) if f.startswith('checkpoint_')) and len(cps) > 3
        for f in ('saved_v.npy', 'saved_logl.npy', 'saved_logwt.npy'):
            assert np.array_equal(np.load(str(tmp_path / 'py' / 'checkpoint' / f)), np.load(str(tmp_path / 'native' / 'checkpoint' / f))), f
        assert open(str(tmp_path / 'py' / 'checkpoint' / cps[-1])).read() == open(str(tmp_path / 'native' / 'checkpoint' / cps[-1])).read()
    else:
        cp = str(tmp_path / 'native' / 'checkpoint')
        last = max(int(f.split('_')[1].split('.')[0]) for f in os.listdir(cp) if f.startswith('checkpoint_'))
        sv, sl = np.load(os.path.join(cp, 'saved_v.npy')), np.load(os.path.join(cp, 'saved_logl.npy'))
        assert len(sl) == last == len(sv) and last > 400          # the files hold the rows before AND after the resume
        assert np.array_equal(sl, b.loglikes[:last]) and np.all(np.diff(sl) >= 0)
        assert abs(b.logz - a.logz) < 0.5


def test_growing_npy_restarts_and_initial_rows(tmp_path):
    """a writer reused for a shorter sequence rewrites its file (it used to append behind the stale rows), and a
    writer created over an existing file with `initial` rows replaces it in one step"""
    from nnest_amd.utils import GrowingNpy
    p = str(tmp_path / 'g.npy')
    g = GrowingNpy(p, ())
    g.sync([1.0, 2.0, 3.0])
    g.sync([])
    g.sync([9.0, 8.0, 7.0, 6.0])
    assert np.array_equal(np.load(p), [9.0, 8.0, 7.0, 6.0])
    np.save(p, np.arange(5.0))                      # a file somebody else wrote (another header length)
    g = GrowingNpy(p, (), initial=[0.0, 1.0])
    assert np.array_equal(np.load(p), [0.0, 1.0]) and not os.path.exists(p + '.tmp')
    g.sync([0.0, 1.0, 2.0])
    assert np.array_equal(np.load(p), [0.0, 1.0, 2.0])
    g2 = GrowingNpy(str(tmp_path / 'v.npy'), (2,), initial=np.ones((3, 2)))
    g2.sync(np.ones((4, 2)))
    assert np.load(str(tmp_path / 'v.npy')).shape == (4, 2)


def test_native_loop_takes_the_first_smallest_live_point():
    """nnest_host_mcmc_consume keeps np.argmin(active_logl) (nested.py:272) in a tournament tree; with MANY tied likelihood values
    the order of the dead points must still be numpy's (the first smallest), through several batches of replacements."""
    import ctypes
    from nnest_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(3)
    N, D, C = 300, 2, 64
    logl = rng.integers(0, 12, N).astype(np.float64)
    u = np.ascontiguousarray(rng.random((N, D)))
    v = u.copy()
    derived = np.zeros((N, 0))
    m_logl, m_v = logl.copy(), v.copy()            # the mirror: the same loop with np.argmin
    cap = 4096
    dead_v, dead_logl, dead_logwt, dead_zprev = np.zeros((cap, D)), np.zeros(cap), np.zeros(cap), np.zeros(cap)
    st = _lib.HostState(logz=-1e300, logvol=0.0, fraction_remain=1.0, max_logl=float(logl.max()), loglstar=0.0, it=1, n_dead=0,
                        accept_point=0, nb=C, first_time=0, resume=_lib.HOST_TOP, worst=0, pad_=0)
    end_u, end_v, end_logl = np.zeros((C, D)), np.zeros((C, D)), np.zeros(C)
    moved, end_d = np.ones(C, dtype=np.uint8), np.zeros((C, 0))
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    expect_v, expect_l = [], []
    m_accept, batches = False, 0
    while True:
        reason = lib.nnest_host_mcmc_consume(ctypes.byref(st), N, D, 0, P(u), P(v), P(logl), P(derived), P(end_u), P(end_v), P(end_logl), P(moved),
                                             P(end_d), C, P(dead_v), P(dead_logl), P(dead_logwt), P(dead_zprev), cap, -1.0, 10 ** 9, 10 ** 9,
                                             10 ** 9)
        assert reason == _lib.HOST_NEED_SAMPLES, reason
        if batches == 12:
            break
        batches += 1
        end_logl[:] = rng.integers(0, 16, C)           # ties among the candidates and with the live points
        end_u[:] = rng.random((C, D))
        end_v[:] = end_u
        moved[:] = rng.random(C) < 0.9
        nb = 0                                         # the mirror consumes the batch
        while True:
            worst = int(np.argmin(m_logl))
            if m_accept:
                expect_v.append(m_v[worst].copy())
                expect_l.append(m_logl[worst])
                m_accept = False
            if nb >= C:
                break
            while nb < C:
                c = nb
                nb += 1
                if moved[c] and end_logl[c] > m_logl[worst]:
                    m_logl[worst], m_v[worst] = end_logl[c], end_v[c]
                    m_accept = True
                    break
        st.nb = 0
        st.resume = _lib.HOST_AFTER_SAMPLES
    n = int(st.n_dead)
    assert n == len(expect_l) and n > 300
    assert np.array_equal(dead_logl[:n], np.array(expect_l))
    assert np.array_equal(dead_v[:n], np.array(expect_v))
    assert np.array_equal(logl, m_logl) and np.array_equal(v, m_v)


def test_background_jobs_keep_the_order_of_the_scalar_rows(tmp_path):
    """round 4: models/netG.pt and the bulk log-Z scalars are written by one worker thread (utils.BackgroundJobs) beside the GPU work.
    scalars.csv must hold the rows in the order of the calls, whichever path a row took, and a job's exception surfaces at wait()."""
    from nnest_amd.utils import BackgroundJobs, ScalarWriter
    jobs = BackgroundJobs()
    w = ScalarWriter(str(tmp_path))
    w.jobs = jobs
    w.add_scalar('logz', -3.0, 1)
    w.add_scalar('logz', -2.5, 2)
    w.add_scalars('logz', np.arange(3, 2003), np.linspace(-2.0, 0.0, 2000))
    w.add_scalar('logz', 0.25, 2003)
    w.add_scalars('logz', [2004, 2005], [0.5, 0.75])
    w.flush()
    rows = open(os.path.join(str(tmp_path), 'scalars.csv')).read().splitlines()
    assert len(rows) == 2005
    assert [int(r.split(',')[1]) for r in rows] == list(range(1, 2006))
    assert rows[0] == 'logz,1,-3.0' and rows[-1] == 'logz,2005,0.75' and rows[2] == 'logz,3,-2.0'
    # the same rows without a worker
    w2 = ScalarWriter(str(tmp_path / 'b'))
    os.makedirs(str(tmp_path / 'b'))
    w2.add_scalar('logz', -3.0, 1)
    w2.add_scalar('logz', -2.5, 2)
    w2.add_scalars('logz', np.arange(3, 2003), np.linspace(-2.0, 0.0, 2000))
    w2.add_scalar('logz', 0.25, 2003)
    w2.add_scalars('logz', [2004, 2005], [0.5, 0.75])
    w2.flush()
    assert open(os.path.join(str(tmp_path / 'b'), 'scalars.csv')).read().splitlines() == rows

    def boom():
        raise OSError('disk full')
    jobs.submit(boom)
    with pytest.raises(OSError):
        jobs.wait()
    jobs.wait()   # reported once


@pytest.mark.parametrize('kw', [dict(volume_switch=0.4), dict(), dict(max_iters=90), dict(volume_switch=0.4, log_interval=7)],
                         ids=['volume_switch', 'efficiency_switch', 'ends_in_the_prior_phase', 'odd_log_interval'])
def test_native_prior_phase_equals_the_python_loop(tmp_path, monkeypatch, kw):
    """Round 6: the 'rejection_prior' phase of run() in the native library (nnest_host_prior_consume: nnest/nested.py:322-334, :362-373
    over Sampler._rejection_prior_sample, nnest/sampler.py:529-543, candidates evaluated a block per launch) against the Python loop
    of nnest_amd/nested.py + nnest_amd/sampler.py on the same seeds -- EQUAL in log Z, H, iterations, likelihood calls, every dead
    point, results.csv and the checkpoint files -- through the hand-over to the MCMC phase.  No GPU here: the likelihood kernel is
    stood in for by the oracle on float32(x) (what the kernel computes), the MCMC batches take the host protocol."""
    import nnest_amd.flow as nflow
    from oracle import oracle as orc  # checker only
    D = 3
    like = Rosenbrock(D)
    monkeypatch.setattr(nflow, 'loglike', lambda like_id, x, scale, device=None, like_params=None:
                        torch.from_numpy(orc.loglike('rosenbrock', np.asarray(x, dtype=np.float32), scale)))

    class S(NestedSampler):
        def _mcmc_endpoints_fused(self, mcmc_steps, step_size, dynamic, init_samples, init_loglikes, loglstar, walker_offset, seed, form=None):
            fid, self._fused_like_id = self._fused_like_id, None     # (the host protocol for the chains: there is no kernel here)
            try:
                s_x, _lat, s_d, s_l, scale, nc = self._mcmc_sample(mcmc_steps, step_size=step_size, dynamic_step_size=dynamic,
                                                                   init_samples=init_samples, init_loglikes=init_loglikes,
                                                                   init_derived=np.empty((init_samples.shape[0], 0)), loglstar=loglstar)
            finally:
                self._fused_like_id = fid
            mv = np.all(s_x[:, 0, :] != s_x[:, -1, :], axis=1)
            return np.concatenate([s_x[:, -1, :], s_l[:, -1:], mv[:, None]], axis=1).astype(np.float64), scale, nc

    def go(native, sub):
        np.random.seed(5)
        torch.manual_seed(5)
        tr = OracleTrainer(D, seed=2)
        tr.netG.device = torch.device('cpu')
        s = S(D, like, transform=lambda x: 5 * x, log_dir=str(tmp_path / sub), num_live_points=100, trainer=tr, log_level=30,
              append_run_num=False, checkpoint_min_seconds=0.0, chain_min_seconds=0.0, native_loop=native, fused=False)
        s._fused_like_id, s._fused_like_params = 0, ()      # as Sampler._fused_eligibility leaves them on a GPU
        assert s._linear_scale == 5.0
        if native:
            assert s._native_prior_ok(['rejection_prior', 'mcmc'], [], None)
        s.run(train_iters=15, mcmc_num_chains=8, mcmc_steps=6, **dict(dict(log_interval=20), **kw))
        return s

    a, b = go(False, 'py'), go(True, 'native')
    assert a.niter > 60
    assert a.logz == b.logz and a.h == b.h and a.niter == b.niter and a.ncall == b.ncall and a.num_retrains == b.num_retrains
    for k in ('samples', 'weights', 'loglikes'):
        assert np.array_equal(getattr(a, k), getattr(b, k)), k
    for name in ('results/results.csv', 'results/final.csv'):
        assert open(str(tmp_path / 'py' / name)).read() == open(str(tmp_path / 'native' / name)).read(), name
    cps = sorted(f for f in os.listdir(str(tmp_path / 'py' / 'checkpoint')) if f.startswith('checkpoint_'))
    assert cps == sorted(f for f in os.listdir(str(tmp_path / 'native' / 'checkpoint')) if f.startswith('checkpoint_')) and len(cps) > 2
    for f in ('saved_v.npy', 'saved_logl.npy', 'saved_logwt.npy'):
        assert np.array_equal(np.load(str(tmp_path / 'py' / 'checkpoint' / f)), np.load(str(tmp_path / 'native' / 'checkpoint' / f))), f
    assert open(str(tmp_path / 'py' / 'checkpoint' / cps[-1])).read() == open(str(tmp_path / 'native' / 'checkpoint' / cps[-1])).read()
