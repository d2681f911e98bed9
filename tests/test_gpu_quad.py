"""GPU parity of the "quad" form of the persistent Metropolis kernel (4 walkers per wave on v_mfma_f32_4x4x1_16B_f32,
nnest_amd/csrc/nnest_quad.hip) and of the batch-wide step-size rule (sampler.py:422-431 over all walkers of a launch;
include/nnest_hip.h NNEST_MH_DYNAMIC_BATCH) against the oracle on the same noise.  The golden-trace checks of the quad
form (the reference's recorded torch noise) are in tests/test_gpu_parity.py::test_mh_trace_vs_golden_recorded_noise."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc  # checker only
from tests.mh_checks import assert_borderline

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def hip():
    from nnest_amd import flow
    if not torch.cuda.is_available():
        pytest.fail('no GPU visible: the -m gpu tests must run on an MI355X')
    return flow


def cpu(t):
    return t.detach().cpu().numpy()


def rel(a, b):
    return float(np.max(np.abs(a - b) / (1.0 + np.abs(b))))


def trained(hip, D=50):
    g = np.load(os.path.join(G, 'mcmc_rosen_d50.npz' if D == 50 else 'flow_d%d.npz' % D))
    w = g['w'] if 'w' in g.files else g['w_trained']
    nvp = hip.HipNVP(D, 16, 3, 1)
    nvp.load_packed(w)
    return nvp, orc.NVP(D, 16, 3, 1, w), g


@pytest.mark.parametrize('C,S', [(1000, 30), (37, 25), (3, 40), (1024, 4), (2000, 12)])   # 2000: two tiles per CU, both nets on one wave
def test_quad_inkernel_noise_vs_oracle_per_walker(hip, C, S):
    """Fixed step size: walkers are independent, so every walker's chain is replayed through the oracle on the kernel's own
    noise (nnest_mh_fill_noise).  A walker whose accept/call counts differ took a borderline decision the other way
    (float32 rounding of a different summation order) and is not compared further; there may be very few."""
    nvp, o, g = trained(hip)
    rng = np.random.RandomState(C)
    init = g['init'][rng.randint(0, g['init'].shape[0], size=C)]
    init_logl = orc.loglike('rosenbrock', init, 5.0)
    loglstar = float(np.min(init_logl)) - 1e3
    step, seed, off = 0.05, 424242, 77
    dz, u = nvp.fill_noise(S, C, seed=seed, walker_offset=off)
    dzc, uc = cpu(dz), cpu(u)
    z, _ = nvp.forward(init)
    logl = torch.from_numpy(init_logl).cuda()
    res = nvp.mh_steps(0, 5.0, z, logl, loglstar, step, S, seed=seed, walker_offset=off, history=True, form='quad')
    hx, hl = cpu(res['hist_x']), cpu(res['hist_logl'])
    margins = np.empty((S, C))
    so, _, lo, _, _, _ = orc.mcmc_sample(o, 'rosenbrock', 5.0, init, init_logl, loglstar, step, False, dzc, uc, margins=margins)
    moved_o = np.sum(np.any(so[:, 1:] != so[:, :-1], axis=2), axis=1)
    same = cpu(res['n_accept']) == moved_o
    assert np.sum(~same) <= max(1, C // 200)
    assert_borderline(hx, so, margins, np.flatnonzero(~same))
    assert rel(hx[same], so[same]) < 5e-5
    assert rel(hl[same], lo[same]) < 5e-5
    assert int(res['n_accept'].sum()) > 0
    # production instantiation (no history): same final state, bit for bit
    z2, _ = nvp.forward(init)
    logl2 = torch.from_numpy(init_logl).cuda()
    res2 = nvp.mh_steps(0, 5.0, z2, logl2, loglstar, step, S, seed=seed, walker_offset=off, form='quad')
    assert torch.equal(z2, z) and torch.equal(logl2, logl) and torch.equal(res2['x'], res['x'])
    assert torch.equal(res2['n_accept'], res['n_accept']) and torch.equal(res2['n_call'], res['n_call'])
    assert rel(cpu(res['x']), hx[:, -1]) == 0.0


@pytest.mark.parametrize('D,like,scale', [(2, 'rosenbrock', 5.0), (20, 'gaussmix', 10.0), (32, 'himmelblau', 5.0),
                                          (100, 'rosenbrock', 5.0), (7, 'gaussian', 3.0), (5, 'shell', 6.0),
                                          (5, 'double_shell', 6.0), (2, 'eggbox', 15.0)])
def test_quad_shapes_and_likelihoods_vs_oracle(hip, D, like, scale):
    """x_dim 2..100 (1, 2 and 4 register groups per class) and every fused likelihood, fixed step, per walker."""
    nvp = hip.HipNVP(D, 16, 3, 1, seed=D)
    o = orc.NVP(D, 16, 3, 1, nvp.store_packed())
    params = {'gaussian': (0.5,), 'shell': (0.1, 2.0, 0.0), 'double_shell': (0.1, 2.0, -1.0, 0.2, 1.5, 1.0)}.get(like)
    C, S = 70, 15
    rng = np.random.RandomState(D)
    init = rng.uniform(-0.5, 0.5, size=(C, D))
    init_logl = orc.loglike(like, init, scale, params)
    dz, u = nvp.fill_noise(S, C, seed=9)
    z, _ = nvp.forward(init)
    logl = torch.from_numpy(init_logl).cuda()
    lid = hip._lib.LIKE_IDS[like]
    res = nvp.mh_steps(lid, scale, z, logl, -1e300, 0.05, S, seed=9, history=True, form='quad', like_params=params)
    hx = cpu(res['hist_x'])
    # the chain under the kernel's own decisions: every stored x must be f^-1 of a latent that moved by the recorded noise
    # (checked through the final state) and every stored logL the oracle's likelihood of the stored x
    lo = orc.loglike(like, hx.reshape(-1, D), scale, params).reshape(C, S + 1)
    hl = cpu(res['hist_logl'])
    assert np.max(np.abs(hl - lo) / (1.0 + np.abs(lo))) < 5e-5
    xo, _ = o.inverse(cpu(z))
    assert rel(cpu(res['x']), xo) < 1e-4
    assert float(res['x'].abs().max()) <= 1.0 and int(res['n_accept'].sum()) > 0
    # with the threshold at -1e300 every in-box proposal that passes the Jacobian test is accepted: calls == accepts
    assert torch.equal(res['n_call'], res['n_accept'])


@pytest.mark.parametrize('C,form,lag', [(1000, 'quad', 0), (1000, 'quad', 2), (333, 'quad', 1), (2000, 'quad', 4), (2000, 'team', 2),
                                        (2000, 'team', 0), (4800, 'reg', 3), (6000, 'image', 2), (20000, 'image', 4)])
def test_batch_wide_step_rule_vs_oracle(hip, C, form, lag):
    """NNEST_MH_DYNAMIC_BATCH: the accept count is taken over the WHOLE launch (the reference's rule at lag 0; with lag L
    the update after step s uses the count of step s - L).  The oracle runs the whole batch with the same lag."""
    nvp, o, g = trained(hip)
    rng = np.random.RandomState(C + lag)
    init = g['init'][rng.randint(0, g['init'].shape[0], size=C)]
    init_logl = orc.loglike('rosenbrock', init, 5.0)
    loglstar = float(np.min(init_logl)) - 1e3
    S, step, seed = 24, 0.3, 99
    dz, u = nvp.fill_noise(S, C, seed=seed)
    z, _ = nvp.forward(init)
    logl = torch.from_numpy(init_logl).cuda()
    res = nvp.mh_steps(0, 5.0, z, logl, loglstar, step, S, dynamic='batch', lag=lag, seed=seed, history=True, form=form)
    hip.HipNVP.check_sync(res)
    margins = np.empty((S, C))
    so, _, lo, sc, ncall, (acc, rej) = orc.mcmc_sample(o, 'rosenbrock', 5.0, init, init_logl, loglstar, step, True, cpu(dz),
                                                       cpu(u), lag=lag, margins=margins)
    scales = cpu(res['scale'])
    assert np.all(scales == scales[0])                       # one rule for the whole batch
    assert abs(float(scales[0]) - sc) < 1e-6 * max(1.0, sc)  # same sequence of majority decisions
    assert sc != step                                        # the rule did act
    moved_o = np.sum(np.any(so[:, 1:] != so[:, :-1], axis=2), axis=1)
    same = cpu(res['n_accept']) == moved_o
    assert np.sum(~same) <= max(1, C // 200)
    assert_borderline(cpu(res['hist_x']), so, margins, np.flatnonzero(~same))
    assert rel(cpu(res['hist_x'])[same], so[same]) < 5e-5
    assert rel(cpu(res['hist_logl'])[same], lo[same]) < 5e-5


def test_batch_rule_at_the_config5_population_on_one_gpu(hip):
    """BASELINE config 5's shape on ONE GPU -- x_dim 100, 8000 walkers: 500 tiles of 16 on 256 CUs, image form, one wave per SIMD
    -- under the reference's batch-wide rule (round 2 refused it there: more workgroups than CUs; the launch shape now
    guarantees residency, two 2-wave workgroups per CU).  Scale sequence and chains against the oracle."""
    D, C, S, lag = 100, 8000, 10, 4
    nvp = hip.HipNVP(D, 16, 3, 1, seed=3)
    o = orc.NVP(D, 16, 3, 1, nvp.store_packed())
    assert nvp.mh_form_for(C, dynamic='batch', lag=lag) == 'image'
    rng = np.random.RandomState(5)
    init = rng.uniform(-0.5, 0.5, size=(C, D))
    init_logl = orc.loglike('rosenbrock', init, 5.0)
    loglstar = float(np.min(init_logl)) - 1e3
    step, seed = 0.05, 123
    dz, u = nvp.fill_noise(S, C, seed=seed)
    z, _ = nvp.forward(init)
    logl = torch.from_numpy(init_logl).cuda()
    res = nvp.mh_steps(0, 5.0, z, logl, loglstar, step, S, dynamic='batch', lag=lag, seed=seed, history=True)
    hip.HipNVP.check_sync(res)
    margins = np.empty((S, C))
    so, _, lo, sc, ncall, _ = orc.mcmc_sample(o, 'rosenbrock', 5.0, init, init_logl, loglstar, step, True, cpu(dz), cpu(u), lag=lag,
                                              margins=margins)
    scales = cpu(res['scale'])
    assert np.all(scales == scales[0]) and abs(float(scales[0]) - sc) < 1e-6 * max(1.0, sc) and sc != step
    moved_o = np.sum(np.any(so[:, 1:] != so[:, :-1], axis=2), axis=1)
    same = cpu(res['n_accept']) == moved_o
    assert np.sum(~same) <= C // 200
    assert_borderline(cpu(res['hist_x']), so, margins, np.flatnonzero(~same))
    assert rel(cpu(res['hist_x'])[same], so[same]) < 5e-5
    # the production build (no history; one wave per SIMD): the same final state bit for bit
    z2, _ = nvp.forward(init)
    logl2 = torch.from_numpy(init_logl).cuda()
    res2 = nvp.mh_steps(0, 5.0, z2, logl2, loglstar, step, S, dynamic='batch', lag=lag, seed=seed)
    hip.HipNVP.check_sync(res2)
    assert torch.equal(z2, z) and torch.equal(res2['n_accept'], res['n_accept']) and torch.equal(res2['scale'], res['scale'])


def test_batch_rule_differs_from_group_rule_and_lag_matters(hip):
    """what the deviation of the per-16-walker rule amounts to, on one launch: different final scales per group, a
    different common scale under the batch rule; lag 0 and lag 2 differ too (both deterministic)."""
    nvp, o, g = trained(hip)
    C, S = 640, 40
    init = g['init'][np.arange(C) % g['init'].shape[0]]
    init_logl = orc.loglike('rosenbrock', init, 5.0)
    out = {}
    for key, kw in {'group': dict(dynamic='group'), 'b0': dict(dynamic='batch', lag=0), 'b2': dict(dynamic='batch', lag=2),
                    'b2again': dict(dynamic='batch', lag=2)}.items():
        z, _ = nvp.forward(init)
        logl = torch.from_numpy(init_logl).cuda()
        res = nvp.mh_steps(0, 5.0, z, logl, float(init_logl.min()) - 1e3, 0.3, S, seed=5, **kw)
        hip.HipNVP.check_sync(res)
        out[key] = (cpu(res['scale']), cpu(z))
    assert len(np.unique(out['group'][0])) > 1
    assert len(np.unique(out['b0'][0])) == 1 and len(np.unique(out['b2'][0])) == 1
    assert np.array_equal(out['b2'][1], out['b2again'][1]) and np.array_equal(out['b2'][0], out['b2again'][0])
    assert not np.array_equal(out['b0'][1], out['b2'][1])


def test_pinned_form_makes_shards_reproduce_the_full_batch(hip):
    """Fixed step: a shard [a, b) launched with walker_offset = a and the form of the full batch pinned equals the slice
    of the full launch bit for bit -- also when the shard alone would have been given another form."""
    nvp, o, g = trained(hip)
    C = 3072  # 192 tiles of 16: the full batch runs the team form (the quad form stops at two 4-walker tiles per CU)
    init = g['init'][np.arange(C) % g['init'].shape[0]]
    init_logl = orc.loglike('rosenbrock', init, 5.0)

    def run(lo, hi, form):
        z, _ = nvp.forward(init[lo:hi])
        logl = torch.from_numpy(init_logl[lo:hi]).cuda()
        nvp.mh_steps(0, 5.0, z, logl, -1e9, 0.03, 20, seed=31, walker_offset=lo, form=form)
        return cpu(z), cpu(logl)

    zf, lf = run(0, C, None)
    zt, lt = run(0, C, 'team')
    assert np.array_equal(zf, zt) and np.array_equal(lf, lt)
    zs, ls = run(512, 1024, 'team')          # 512 walkers alone would run the quad form
    assert np.array_equal(zs, zf[512:1024]) and np.array_equal(ls, lf[512:1024])
    zq, lq = run(512, 1024, None)            # the quad form: the same chains to rounding (another summation order)
    assert rel(zq, zs) < 1e-5 and rel(lq, ls) < 1e-5


def test_batch_rule_refused_when_the_grid_may_not_be_resident(hip):
    nvp, o, g = trained(hip)
    C = 70000
    z = torch.zeros(C, 50, device='cuda')
    logl = torch.zeros(C, dtype=torch.float64, device='cuda')
    with pytest.raises(hip._lib.NnestHipError):
        nvp.mh_steps(0, 5.0, z, logl, -1e9, 0.03, 2, dynamic='batch')
    with pytest.raises(hip._lib.NnestHipError):
        nvp.mh_steps(0, 5.0, z[:5000].contiguous(), logl[:5000].contiguous(), -1e9, 0.03, 2, form='quad')
