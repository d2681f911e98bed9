"""GPU parity of the "solo" form of the persistent Metropolis kernel (ONE walker per wave; layers as v_fmac_f32 with DPP row
rotations, both nets of a coupling block in the two halves of the wave: nnest_amd/csrc/nnest_solo.hip) against the oracle
on the same noise -- Sampler._mcmc_sample (nnest/sampler.py:229-463), hard-constraint and unconstrained branch, fixed step
and the batch-wide step rule (sampler.py:422-431) relayed by the noise wave.  The form is what BASELINE config 2's 1000
walkers run by default."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc  # checker only
from tests.mh_checks import assert_borderline

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def trained(hip):
    g = np.load(os.path.join(G, 'mcmc_rosen_d50.npz'))
    nvp = hip.HipNVP(50, 16, 3, 1)
    nvp.load_packed(g['w'])
    return nvp, orc.NVP(50, 16, 3, 1, g['w']), g


def test_solo_is_what_config_2_runs(hip):
    """the library's own answer (nnest_mh_form_for): 1000 walkers at x_dim 50 run the solo form under a fixed step and under
    the batch-wide rule at the default lag and at lag 0 (every step exact); lag 1 / 2 (no time for the relay), the per-group rule and
    populations beyond four walkers per CU go to the other forms"""
    nvp, _, _ = trained(hip)
    assert nvp.mh_form_for(1000) == 'solo' and nvp.mh_form_for(1000, dynamic='batch') == 'solo'
    # round 5: lag 0 -- the reference's rule itself -- is a solo schedule too (every step an exact step); lag 1 / 2 stay with the quad form
    assert nvp.mh_form_for(1000, dynamic='batch', lag=0) == 'solo' and nvp.mh_form_for(1000, dynamic='batch', lag=2) == 'quad' and nvp.mh_form_for(1000, dynamic='group') in ('team', 'reg', 'image')
    # round 4: two / three walkers per SIMD (8 / 12 net waves per workgroup, weights in LDS) carry the form to 3060 walkers
    assert nvp.mh_form_for(2000) == 'solo' and nvp.mh_form_for(3000) == 'solo' and nvp.mh_form_for(3061) != 'solo'
    assert nvp.mh_form_for(4000) == 'team' and nvp.mh_form_for(100000) == 'image'
    assert nvp.mh_form_for(100000, dynamic='batch') is None                    # grid may not be resident: refused
    assert nvp.mh_form_for(1000, form='team') == 'team' and nvp.mh_form_for(5000, form='solo') is None
    big = hip.HipNVP(100, 16, 3, 1, seed=0)
    assert big.mh_form_for(1000) == 'solo' and big.mh_form_for(8000) == 'image' and big.mh_form_for(8000, form='reg') is None   # x_dim 97..128: weights in LDS
    mid = hip.HipNVP(80, 16, 3, 1, seed=0)
    assert mid.mh_form_for(1000) == 'solo' and mid.mh_form_for(1000, dynamic='batch', lag=0) == 'solo' and mid.mh_form_for(1000, dynamic='batch', lag=1) == 'quad'   # x_dim 65..96: three slots per class
    wide = hip.HipNVP(20, 32, 3, 1, seed=0)
    assert wide.mh_form_for(500) == 'image' and wide.mh_form_for(500, form='quad') is None


@pytest.mark.parametrize('C,S', [(1000, 30), (37, 25), (3, 40), (1020, 4), (1, 50)])
def test_solo_inkernel_noise_vs_oracle_per_walker(hip, C, S):
    """Fixed step size: walkers are independent, so every walker's chain is replayed through the oracle on the kernel's own
    noise (nnest_mh_fill_noise).  A walker whose accept count differs took a borderline decision the other way (float32
    rounding of a different summation order): asserted to be borderline, then left out; there may be very few."""
    nvp, o, g = trained(hip)
    rng = np.random.RandomState(C)
    init = g['init'][rng.randint(0, g['init'].shape[0], size=C)]
    init_logl = orc.loglike('rosenbrock', init, 5.0)
    loglstar = float(np.min(init_logl)) - 1e3
    step, seed, off = 0.05, 424242, 77
    dz, u = nvp.fill_noise(S, C, seed=seed, walker_offset=off)
    z, _ = nvp.forward(init)
    logl = torch.from_numpy(init_logl).cuda()
    res = nvp.mh_steps(0, 5.0, z, logl, loglstar, step, S, seed=seed, walker_offset=off, history=True, form='solo')
    hx, hl = cpu(res['hist_x']), cpu(res['hist_logl'])
    margins = np.empty((S, C))
    so, _, lo, _, ncall, _ = orc.mcmc_sample(o, 'rosenbrock', 5.0, init, init_logl, loglstar, step, False, cpu(dz), cpu(u),
                                             margins=margins)
    moved_o = np.sum(np.any(so[:, 1:] != so[:, :-1], axis=2), axis=1)
    same = cpu(res['n_accept']) == moved_o
    assert np.sum(~same) <= max(1, C // 200)
    assert_borderline(hx, so, margins, np.flatnonzero(~same))
    assert rel(hx[same], so[same]) < 5e-5
    assert rel(hl[same], lo[same]) < 5e-5
    assert int(res['n_accept'].sum()) > 0
    if same.all():
        assert int(res['n_call'].sum()) == ncall
    # production instantiation (no history): same final state, bit for bit; and the default form IS this one
    for form in ('solo', None):
        z2, _ = nvp.forward(init)
        logl2 = torch.from_numpy(init_logl).cuda()
        res2 = nvp.mh_steps(0, 5.0, z2, logl2, loglstar, step, S, seed=seed, walker_offset=off, form=form)
        assert torch.equal(z2, z) and torch.equal(logl2, logl) and torch.equal(res2['x'], res['x'])
        assert torch.equal(res2['n_accept'], res['n_accept']) and torch.equal(res2['n_call'], res['n_call'])
    assert rel(cpu(res['x']), hx[:, -1]) == 0.0


@pytest.mark.parametrize('D,like,scale', [(2, 'rosenbrock', 5.0), (3, 'rosenbrock', 5.0), (20, 'gaussmix', 10.0), (32, 'himmelblau', 5.0),
                                          (33, 'rosenbrock', 5.0), (50, 'rosenbrock', 5.0), (64, 'rosenbrock', 5.0),
                                          (65, 'rosenbrock', 5.0), (80, 'rosenbrock', 5.0), (96, 'rosenbrock', 5.0), (81, 'gaussmix', 10.0),
                                          (97, 'rosenbrock', 5.0), (100, 'rosenbrock', 5.0), (128, 'rosenbrock', 5.0), (100, 'gaussmix', 10.0),
                                          (7, 'gaussian', 3.0), (5, 'shell', 6.0), (5, 'double_shell', 6.0), (2, 'eggbox', 15.0)])
def test_solo_shapes_and_likelihoods_vs_oracle(hip, D, like, scale):
    """x_dim 2..64 (one and two register groups per class, odd sizes, full tiles) and every fused likelihood, fixed step,
    per walker against the oracle's chain."""
    nvp = hip.HipNVP(D, 16, 3, 1, seed=D)
    o = orc.NVP(D, 16, 3, 1, nvp.store_packed())
    params = {'gaussian': (0.5,), 'shell': (0.1, 2.0, 0.0), 'double_shell': (0.1, 2.0, -1.0, 0.2, 1.5, 1.0)}.get(like)
    C, S = 70, 15
    rng = np.random.RandomState(D)
    init = rng.uniform(-0.5, 0.5, size=(C, D))
    init_logl = orc.loglike(like, init, scale, params)
    z, _ = nvp.forward(init)
    logl = torch.from_numpy(init_logl).cuda()
    lid = hip._lib.LIKE_IDS[like]
    res = nvp.mh_steps(lid, scale, z, logl, -1e300, 0.05, S, seed=9, history=True, form='solo', like_params=params)
    hx = cpu(res['hist_x'])
    lo = orc.loglike(like, hx.reshape(-1, D), scale, params).reshape(C, S + 1)   # every stored logL belongs to its stored x
    hl = cpu(res['hist_logl'])
    assert np.max(np.abs(hl - lo) / (1.0 + np.abs(lo))) < 5e-5
    xo, _ = o.inverse(cpu(z))                                                    # the final x is f^-1 of the final latent
    assert rel(cpu(res['x']), xo) < 1e-4
    assert float(res['x'].abs().max()) <= 1.0 and int(res['n_accept'].sum()) > 0
    assert torch.equal(res['n_call'], res['n_accept'])   # threshold -1e300: every pre-accepted proposal is accepted
    if like == 'rosenbrock':   # and the whole chain against the oracle's on the same noise
        dz, u = nvp.fill_noise(S, C, seed=9)
        margins = np.empty((S, C))
        so, _, _, _, _, _ = orc.mcmc_sample(o, like, scale, init, init_logl, -1e300, 0.05, False, cpu(dz), cpu(u), margins=margins)
        moved_o = np.sum(np.any(so[:, 1:] != so[:, :-1], axis=2), axis=1)
        same = cpu(res['n_accept']) == moved_o
        assert np.sum(~same) <= 1
        assert_borderline(hx, so, margins, np.flatnonzero(~same))
        assert rel(hx[same], so[same]) < 5e-5


@pytest.mark.parametrize('C,lag,warm', [(1000, 3, 0), (1000, 4, 0), (333, 5, 0), (1017, 8, 0), (64, 15, 0), (1000, 8, 16), (333, 3, 1),
                                        (1017, 8, 39), (500, 5, 200), (64, 15, 7), (1000, 0, 0), (333, 0, 0), (5, 0, 0)])
def test_solo_batch_wide_step_rule_vs_oracle(hip, C, lag, warm):
    """NNEST_MH_DYNAMIC_BATCH relayed by the noise wave: the accept count is taken over the WHOLE launch, `lag` steps behind --
    after `warm` steps under the exact rule (NNEST_MH_WARM; the product's default is 16 in front of lag 8).  lag 0 (round 5) is the
    reference's rule itself, sampler.py:422-431: every step exact, the last step's vote in scale_out.
    The oracle runs the whole batch with the same schedule: same scale sequence, same chains."""
    nvp, o, g = trained(hip)
    rng = np.random.RandomState(C + lag)
    init = g['init'][rng.randint(0, g['init'].shape[0], size=C)]
    init_logl = orc.loglike('rosenbrock', init, 5.0)
    loglstar = float(np.min(init_logl)) - 1e3
    S, step, seed = 40, 0.3, 99
    dz, u = nvp.fill_noise(S, C, seed=seed)
    z, _ = nvp.forward(init)
    logl = torch.from_numpy(init_logl).cuda()
    res = nvp.mh_steps(0, 5.0, z, logl, loglstar, step, S, dynamic='batch', lag=lag, seed=seed, history=True, form='solo', warm=warm)
    hip.HipNVP.check_sync(res)
    margins = np.empty((S, C))
    so, _, lo, sc, ncall, (acc, rej) = orc.mcmc_sample(o, 'rosenbrock', 5.0, init, init_logl, loglstar, step, True, cpu(dz),
                                                       cpu(u), lag=lag, margins=margins, warm=warm)
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


@pytest.mark.parametrize('D,C,lag,warm', [(100, 500, 8, 16), (128, 1000, 4, 0), (80, 333, 8, 16)])
def test_solo_lds_weights_batch_rule_vs_oracle(hip, D, C, lag, warm):
    """x_dim 65..128 (the lane's weights in LDS, three or four slots per class): BASELINE config 5's per-GPU population under the product's
    step rule, scale sequence and chains against the oracle"""
    nvp = hip.HipNVP(D, 16, 3, 1, seed=D)
    o = orc.NVP(D, 16, 3, 1, nvp.store_packed())
    assert nvp.mh_form_for(C, dynamic='batch', lag=lag, warm=warm) == 'solo'
    rng = np.random.RandomState(C)
    init = rng.uniform(-0.5, 0.5, size=(C, D))
    init_logl = orc.loglike('rosenbrock', init, 5.0)
    loglstar = float(np.min(init_logl)) - 1e3
    S, step, seed = 40, 0.05, 7
    dz, u = nvp.fill_noise(S, C, seed=seed)
    z, _ = nvp.forward(init)
    logl = torch.from_numpy(init_logl).cuda()
    res = nvp.mh_steps(0, 5.0, z, logl, loglstar, step, S, dynamic='batch', lag=lag, warm=warm, seed=seed, history=True)
    hip.HipNVP.check_sync(res)
    margins = np.empty((S, C))
    so, _, lo, sc, ncall, _ = orc.mcmc_sample(o, 'rosenbrock', 5.0, init, init_logl, loglstar, step, True, cpu(dz), cpu(u), lag=lag,
                                              margins=margins, warm=warm)
    scales = cpu(res['scale'])
    assert np.all(scales == scales[0]) and abs(float(scales[0]) - sc) < 1e-6 * max(1.0, sc) and sc != step
    moved_o = np.sum(np.any(so[:, 1:] != so[:, :-1], axis=2), axis=1)
    same = cpu(res['n_accept']) == moved_o
    assert np.sum(~same) <= max(1, C // 200)
    assert_borderline(cpu(res['hist_x']), so, margins, np.flatnonzero(~same))
    assert rel(cpu(res['hist_x'])[same], so[same]) < 5e-5
    assert rel(cpu(res['hist_logl'])[same], lo[same]) < 5e-5
    z2, _ = nvp.forward(init)          # production instantiation (no history): the same final state bit for bit
    logl2 = torch.from_numpy(init_logl).cuda()
    res2 = nvp.mh_steps(0, 5.0, z2, logl2, loglstar, step, S, dynamic='batch', lag=lag, warm=warm, seed=seed)
    assert torch.equal(z2, z) and torch.equal(res2['n_accept'], res['n_accept']) and torch.equal(res2['scale'], res['scale'])


def test_solo_agrees_with_the_quad_form_and_shards(hip):
    """the same launch in the solo and the quad form: the same chains to rounding; a shard with walker_offset reproduces the
    slice of the full solo launch bit for bit (a walker's chain depends on its own noise stream only)"""
    nvp, o, g = trained(hip)
    C = 600
    init = g['init'][np.arange(C) % g['init'].shape[0]]
    init_logl = orc.loglike('rosenbrock', init, 5.0)

    def run(lo, hi, form, **kw):
        z, _ = nvp.forward(init[lo:hi])
        logl = torch.from_numpy(init_logl[lo:hi]).cuda()
        res = nvp.mh_steps(0, 5.0, z, logl, -1e9, 0.03, 25, seed=31, walker_offset=lo, form=form, **kw)
        return cpu(z), cpu(logl), cpu(res['n_accept'])

    zs, ls, ns = run(0, C, 'solo')
    zq, lq, nq = run(0, C, 'quad')
    same = ns == nq
    assert np.sum(~same) <= 3
    assert rel(zs[same], zq[same]) < 2e-5 and rel(ls[same], lq[same]) < 2e-5
    zp, lp, _ = run(201, 333, 'solo')
    assert np.array_equal(zp, zs[201:333]) and np.array_equal(lp, ls[201:333])


def test_solo_unconstrained_branch_vs_oracle(hip):
    """loglstar = None (sampler.py:371-410): likelihood and box prior in the Metropolis ratio"""
    nvp, o, g = trained(hip)
    C, S = 200, 20
    init = g['init'][np.arange(C) % g['init'].shape[0]]
    init_logl = orc.loglike('rosenbrock', init, 5.0)
    dz, u = nvp.fill_noise(S, C, seed=3)
    z, _ = nvp.forward(init)
    logl = torch.from_numpy(init_logl).cuda()
    res = nvp.mh_steps(0, 5.0, z, logl, None, 0.01, S, seed=3, history=True, form='solo')
    so, _, lo, _, ncall, (acc, _) = orc.mcmc_sample(o, 'rosenbrock', 5.0, init, init_logl, float('nan'), 0.01, False, cpu(dz), cpu(u))
    moved_o = np.sum(np.any(so[:, 1:] != so[:, :-1], axis=2), axis=1)
    same = cpu(res['n_accept']) == moved_o
    assert np.sum(~same) <= 3 and int(res['n_call'].sum()) == C * S == ncall
    assert rel(cpu(res['hist_x'])[same], so[same]) < 5e-5
    assert rel(cpu(res['hist_logl'])[same], lo[same]) < 2e-4


def test_solo_golden_trace_recorded_noise(hip):
    """the reference's own recorded torch noise (tests/golden/mcmc_rosen_d50.npz, fixed step): every decision and state"""
    g = np.load(os.path.join(G, 'mcmc_rosen_d50.npz'))
    assert not bool(g['dynamic'])
    nvp = hip.HipNVP(50, 16, 3, 1)
    nvp.load_packed(g['w'])
    S, C, _ = g['dz'].shape
    z, _ = nvp.forward(g['init'])
    logl = torch.from_numpy(g['init_logl']).cuda().contiguous()
    res = nvp.mh_steps(0, float(g['scale']), z, logl, float(g['loglstar']), float(g['step']), S,
                       noise=(torch.from_numpy(g['dz']), torch.from_numpy(g['u'])), history=True, form='solo')
    assert int(res['n_call'].sum()) == int(g['ncall']) and int(res['n_accept'].sum()) == int(g['total_accepted'])
    assert rel(cpu(res['hist_x']), g['samples']) < 3e-5 and rel(cpu(res['hist_logl']), g['loglikes']) < 3e-5
    assert rel(cpu(z), g['latent'][:, -1]) < 3e-5


def test_solo_golden_dynamic_trace_recorded_noise_under_the_exact_rule(hip):
    """the reference's recorded trace WITH its step-size adaptation on (tests/golden/mcmc_rosen_d50_dyn.npz: torch's noise, every
    state, the final scale) through the solo form at lag 0 -- the rule of sampler.py:422-431 applied over the whole batch, every
    step: each decision, each state, the scale the reference ends with (round 4 could only run this trace on the quad form)."""
    g = np.load(os.path.join(G, 'mcmc_rosen_d50_dyn.npz'))
    assert bool(g['dynamic'])
    nvp = hip.HipNVP(50, 16, 3, 1)
    nvp.load_packed(g['w'])
    S, C, _ = g['dz'].shape
    assert nvp.mh_form_for(C, dynamic='batch', lag=0) == 'solo'
    z, _ = nvp.forward(g['init'])
    logl = torch.from_numpy(g['init_logl']).cuda().contiguous()
    res = nvp.mh_steps(0, float(g['scale']), z, logl, float(g['loglstar']), float(g['step']), S, dynamic='batch', lag=0,
                       noise=(torch.from_numpy(g['dz']), torch.from_numpy(g['u'])), history=True, form='solo')
    hip.HipNVP.check_sync(res)
    assert int(res['n_call'].sum()) == int(g['ncall']) and int(res['n_accept'].sum()) == int(g['total_accepted'])
    assert rel(cpu(res['hist_x']), g['samples']) < 3e-5 and rel(cpu(res['hist_logl']), g['loglikes']) < 3e-5
    assert rel(cpu(z), g['latent'][:, -1]) < 3e-5
    assert abs(float(res['scale'][0]) - float(g['scale_out'])) < 1e-6 * max(1.0, float(g['scale_out']))


@pytest.mark.parametrize('C', [1000, 2000, 3000])
def test_eight_and_twelve_walkers_per_workgroup_run_the_same_chains(hip, C):
    """Round 4: beyond 4 walkers per compute unit the solo form puts 8 / 12 net waves in a workgroup and reads its weights from
    LDS (nnest_solo.hip: solo_walkers_per_group).  A walker's chain must not depend on how many walkers share its workgroup:
    the first 1000 walkers of a 2000- / 3000-walker launch under a fixed step are, bit for bit, the 1000-walker launch (4 per
    workgroup, weights in registers), and every walker is replayed through the oracle on the kernel's own noise."""
    nvp, o, g = trained(hip)
    assert nvp.mh_form_for(C) == 'solo'
    rng = np.random.RandomState(5)
    init = g['init'][rng.randint(0, g['init'].shape[0], size=C)]
    z0, _ = nvp.forward(init)
    l0 = hip.loglike(0, init, 5.0)
    star, S = float(np.median(cpu(l0))) - 50.0, 40
    out = {}
    for n in (1000, C):
        z, l = z0[:n].clone(), l0[:n].clone()
        res = nvp.mh_steps(0, 5.0, z, l, star, 0.05, S, seed=11)
        out[n] = (cpu(z), cpu(l), cpu(res['n_accept']), cpu(res['n_call']), cpu(res['x']))
    for a, b in zip(out[1000], out[C]):
        assert np.array_equal(a, b[:1000])
    # and under the batch-wide rule (exact steps, then lagged) the larger launch follows the oracle's replay of its own noise
    init_logl = orc.loglike('rosenbrock', init, 5.0)
    z = z0.clone()
    l = torch.from_numpy(init_logl).cuda()
    res = nvp.mh_steps(0, 5.0, z, l, star, 0.3, S, seed=12, dynamic='batch', lag=3, warm=4)
    nvp.check_sync(res)
    dz, u = nvp.fill_noise(S, C, seed=12)
    margins = np.empty((S, C))
    so, _, lo, sc, ncall, _ = orc.mcmc_sample(o, 'rosenbrock', 5.0, init, init_logl, star, 0.3, True, cpu(dz), cpu(u), lag=3, margins=margins, warm=4)
    scales = cpu(res['scale'])
    assert np.all(scales == scales[0]) and abs(float(scales[0]) - sc) < 1e-6 * max(1.0, sc) and sc != 0.3
    moved_o = np.sum(np.any(so[:, 1:] != so[:, :-1], axis=2), axis=1)
    same = cpu(res['n_accept']) == moved_o
    assert np.sum(~same) <= max(1, C // 200)
    assert rel(cpu(res['x'])[same], so[same, -1]) < 5e-5


@pytest.mark.parametrize('form', ['solo', 'quad', 'team', 'reg', 'image'])
def test_usable_chain_flag_is_the_references_test(hip, form):
    """nested.py:432: a chain's end may replace a live point only if EVERY coordinate of its last x differs from its first
    (np.all(samples[:, 0] != samples[:, -1])).  Every K4 form evaluates exactly that at the end of the launch against the first x it
    computed (NNEST_MH_ALL_MOVED in the accept count's word; round 4 took "accepted at least once" for it -- a stated deviation,
    now gone).  Checked against the launch's own history, at a late-run proposal scale and at scales so small that an accepted
    move leaves some float32 coordinates where they were -- where the two notions part."""
    nvp, o, g = trained(hip)
    C = 600
    parted = 0
    for step in (0.05, 1e-5, 2e-7):
        rng = np.random.RandomState(17)
        init = g['init'][rng.randint(0, g['init'].shape[0], size=C)]
        z, _ = nvp.forward(init)
        l = hip.loglike(0, init, 5.0)
        star = float(np.min(cpu(l))) - 1e3
        res = nvp.mh_steps(0, 5.0, z, l, star, step, 12, seed=3, history=True, form=form)
        hx, na = cpu(res['hist_x']), cpu(res['n_accept'])
        assert np.all((na >= 0) & (na <= 12))                                  # the flag does not leak into the count
        moved_ref = np.all(hx[:, 0] != hx[:, -1], axis=1)
        assert np.array_equal(cpu(res['moved']), moved_ref), (form, step, int(np.sum(cpu(res['moved']) != moved_ref)))
        assert not np.any(moved_ref & (na == 0))                               # no move without an accept
        parted += int(np.sum((na > 0) & ~moved_ref))
    assert parted > 0                                                          # the small scales did separate the two notions


@pytest.mark.gpu
def test_every_form_writes_both_counters_of_every_walker(hip, monkeypatch):
    """The driver allocates n_accept / n_call uninitialised (flow.py: no fill launches in front of the kernel), so a form that
    left a walker's counters unwritten would feed garbage into the usable-chain flag and the call accounting.  Here
    every int32 allocation is poisoned first, at ragged populations that leave every form's last tile / workgroup partly empty."""
    import torch
    nvp, o, g = trained(hip)
    orig = torch.empty

    def poisoned(*a, **k):
        t = orig(*a, **k)
        if t.dtype == torch.int32:
            t.fill_(0x7fffffff)
        return t

    forms = set()
    for C, dyn in ((1, False), (37, False), (37, 'batch'), (131, True), (1001, 'batch'), (1021, False), (2003, 'batch'), (3059, 'batch'),
                   (4001, 'batch'), (4001, True), (9001, False)):
        rng = np.random.RandomState(C)
        init = g['init'][rng.randint(0, g['init'].shape[0], size=C)]
        z0, _ = nvp.forward(init)
        l0 = hip.loglike(0, init, 5.0)
        star = float(np.median(cpu(l0))) if C > 1 else float(cpu(l0)[0]) - 1.0
        monkeypatch.setattr(torch, 'empty', poisoned)
        try:
            res = nvp.mh_steps(0, 5.0, z0.clone(), l0.clone(), star, 0.02, 120, seed=C, dynamic=dyn)
        finally:
            monkeypatch.setattr(torch, 'empty', orig)
        if dyn == 'batch':
            nvp.check_sync(res)
        na, nc = cpu(res['n_accept']), cpu(res['n_call'])
        assert np.all((na >= 0) & (na <= 120) & (nc >= na) & (nc <= 120)), (C, dyn, int(na.max()), int(nc.max()))
        lag = nvp.default_lag(C) if dyn == 'batch' else 0
        forms.add(nvp.mh_form_for(C, dynamic=dyn, lag=lag, warm=nvp.default_warm(C, dyn, lag) if lag else 0))
    assert len(forms) >= 3, forms     # solo, quad / team / image ... all went through it


def test_exact_steps_reading_the_counters_themselves_give_the_window_words_chains(hip, tmp_path):
    """Round 5: an exact step's vote is read from the tiles' counters directly (one memory-side hop less than the window word the
    publisher workgroup writes).  Same counters, same rule (2 x accepted > walkers): the chains under lag 0 and under the product's
    schedule are the ones NNEST_SOLO_VOTE=window (the round-4 path) produces, bit for bit."""
    import subprocess
    import sys
    code = '''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from nnest_amd import flow
nvp = flow.HipNVP(50, 16, 3, 1, seed=0)
u0 = np.random.RandomState(0).uniform(-1, 1, size=(1000, 50))
z0, _ = nvp.forward(u0)
l0 = flow.loglike(0, u0, 5.0)
out = {}
for name, kw in (('lag0', dict(dynamic='batch', lag=0)), ('product', dict(dynamic='batch'))):
    z, l = z0.clone(), l0.clone()
    res = nvp.mh_steps(0, 5.0, z, l, float(l0.min()), 1 / np.sqrt(50), 60, seed=11, form='solo', **kw)
    flow.HipNVP.check_sync(res)
    out[name + '_x'] = res['x'].cpu().numpy(); out[name + '_a'] = res['n_accept'].cpu().numpy(); out[name + '_s'] = res['scale'].cpu().numpy()
np.savez(sys.argv[1], **out)
''' % ROOT
    outs = {}
    for mode in ('direct', 'window'):
        p = str(tmp_path / (mode + '.npz'))
        env = dict(os.environ)
        env.pop('NNEST_SOLO_VOTE', None)
        if mode == 'window':
            env['NNEST_SOLO_VOTE'] = 'window'
        r = subprocess.run([sys.executable, '-c', code, p], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[mode] = np.load(p)
    for k in outs['direct'].files:
        assert np.array_equal(outs['direct'][k], outs['window'][k]), k
