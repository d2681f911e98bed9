"""Shared check of the GPU Metropolis tests: walkers whose chain differs from the oracle's must have taken a
rounding-borderline decision.

The kernels sum in another order than the oracle (K split over lanes, MFMA accumulation), so a proposal whose Jacobian
ratio sits within float32 rounding of `u`, whose likelihood sits within rounding of L*, or whose coordinate sits within
rounding of the box edge can be decided the other way; from that step on the two chains are different chains.  The tests
exclude such walkers from the element-wise comparison -- and assert here that the exclusion is what it claims to be: at the
FIRST step where the kernel's chain leaves the oracle's, the oracle's decision margin (orc_set_margin_out,
oracle/nnest_oracle.c: min of |u - ratio|, |logL' - L*| / (1 + |L*|), ||x'_d| - 1|) is at rounding level."""
import numpy as np

# float32: log-det sums of ~75 terms of magnitude <~ 50 differ by a few ulp (~1e-5) between summation orders, the ratio
# e^dlogdet inherits that relative error; the likelihood is a float64 sum of float32 terms of a float32 x' that itself
# differs by ~1e-6 relative -- on Rosenbrock's ridge (100 (x_{i+1} - x_i^2)^2) that is ~1e-5 of logL
MARGIN_TOL = 1e-4


def first_divergence(h_gpu, h_orc, tol=1e-3):
    """per walker: first step index s >= 1 where the chains differ by more than tol (relative), or -1"""
    d = np.max(np.abs(h_gpu - h_orc) / (1.0 + np.abs(h_orc)), axis=2) > tol      # [C, S+1]
    first = np.where(d.any(axis=1), d.argmax(axis=1), -1)
    return first


def assert_borderline(h_gpu, h_orc, margins, walkers, tol=MARGIN_TOL):
    """walkers: indices excluded from the element-wise comparison.  Each must leave the oracle's chain at a step the oracle
    decided within `tol` of a threshold."""
    first = first_divergence(h_gpu[walkers], h_orc[walkers])
    for w, s in zip(np.asarray(walkers), first):
        assert s >= 1, 'walker %d was excluded but its chain equals the oracle\'s' % w
        assert margins[s - 1, w] < tol, ('walker %d leaves the oracle chain at step %d where the oracle\'s decision margin is %.3g '
                                         '(not a rounding-borderline decision)' % (w, s, margins[s - 1, w]))


def spline_mcmc_trace(o, z0, logl0, loglstar, step, dz, u, adapt=True, like='rosenbrock', like_scale=5.0, margins=None):
    """Sampler._mcmc_sample, hard-constraint branch (nnest/sampler.py:291-444), one adaptation group (16 walkers under the per-group
    rule, the whole batch under the reference's own; adapt=False: a fixed step), the flow evaluated by the oracle object `o`
    (orc.Spline: nnest/networks.py:458-556 behind ActNorm + 1x1 conv): returns the per-step x / z / logl histories and the counters.
    margins [S, C] (optional, filled): the decision margin of every step and walker -- min of |u - ratio|,
    |logL' - L*| / (1 + |L*|) for rows that passed the first test, and the distance of the nearest coordinate from the box edge."""
    from oracle import oracle as orc  # checker only
    S, C, D = dz.shape
    z = z0.astype(np.float32).copy()
    x, ld = o.inverse(z)
    logl = logl0.astype(np.float64).copy()
    hx, hz, hl = [x.copy()], [z.copy()], [logl.copy()]
    scale = float(step)
    accept = reject = ncall = nacc = 0
    for it in range(S):
        zp = (z + dz[it] * np.float32(scale)).astype(np.float32)
        xp, ldp = o.inverse(zp)
        log_ratio = (ldp - ld).astype(np.float32)
        inbox = orc.prior_inbox(xp) == 0
        log_ratio[~inbox] = -np.inf
        with np.errstate(over='ignore'):
            ratio = np.minimum(np.exp(log_ratio), np.float32(1.0))
        pre = u[it] < ratio
        lp = orc.loglike(like, xp, like_scale)
        acc = pre & np.isfinite(lp) & (lp > loglstar)
        if margins is not None:
            m = np.abs(u[it].astype(np.float64) - ratio.astype(np.float64))
            m = np.minimum(m, np.min(np.abs(np.abs(xp.astype(np.float64)) - 1.0), axis=1))
            m = np.where(pre, np.minimum(m, np.abs(lp - loglstar) / (1.0 + abs(loglstar))), m)
            margins[it] = m
        ncall += int(pre.sum())
        nacc += int(acc.sum())
        z[acc] = zp[acc]; x[acc] = xp[acc]; ld[acc] = ldp[acc]; logl[acc] = lp[acc]
        if adapt:
            if 2 * int(acc.sum()) > C:
                accept += 1
            else:
                reject += 1
            if accept > reject:
                scale *= np.exp(1. / (1 + accept))
            if accept < reject:
                scale /= np.exp(1. / (1 + reject))
        hx.append(x.copy()); hz.append(z.copy()); hl.append(logl.copy())
    return dict(x=np.stack(hx, 1), z=np.stack(hz, 1), logl=np.stack(hl, 1), ncall=ncall, nacc=nacc, scale=scale)


def spline_fixture_launch(path, rule, reps=1):
    """One launch of the spline proposal kernel (nnest_spline_mh_constrained_steps) over a reference-recorded trace
    (tests/golden/mcmc_spline_*.npz): the reference's noise replayed, histories on.  rule: 'fixed' | 'batch' (the reference's rule
    over the whole batch, lag 0) | 'group' (the same rule per 16 walkers).  reps > 1: the fixture's walkers (a multiple of 16) and
    their noise repeated `reps` times -- every 16-walker group is then the reference's batch again (large populations reach the
    one-wave-per-tile form).  Returns numpy arrays + the form the library says it ran."""
    import torch
    from nnest_amd.spline import HipSpline
    g = np.load(path)
    D, S, C = int(g['D']), g['dz'].shape[0], g['dz'].shape[1]
    sp = HipSpline(D, int(g['H']), int(g['B']), int(g['K']), float(g['tail']))
    sp.load_packed(g['w'], g['P'])
    sp.data_dep_init_done = True
    init, init_logl, dz, u = g['init'], g['init_logl'], g['dz'], g['u']
    if reps > 1:
        assert C % 16 == 0
        init, init_logl = np.tile(init, (reps, 1)), np.tile(init_logl, reps)
        dz, u = np.tile(dz, (1, reps, 1)), np.tile(u, (1, reps))
    z, _ = sp.forward(init)                                   # sampler.py:264
    z0 = z.detach().cpu().numpy().copy()
    logl = torch.from_numpy(np.ascontiguousarray(init_logl)).cuda()
    kw = {'fixed': dict(dynamic=False), 'batch': dict(dynamic='batch', lag=0), 'group': dict(dynamic='group')}[rule]
    form = sp.kernel_form_for(C * reps, **kw)
    res = sp.mh_steps(0, float(g['scale']), z, logl, float(g['loglstar']), float(g['step']), S,
                      noise=(torch.from_numpy(np.ascontiguousarray(dz)), torch.from_numpy(np.ascontiguousarray(u))), history=True, **kw)
    HipSpline.check_sync(res)
    c = lambda t: t.detach().cpu().numpy()
    return dict(form=form, z0=z0, z=c(z), logl=c(logl), x=c(res['x']), hist_x=c(res['hist_x']), hist_logl=c(res['hist_logl']),
                n_accept=c(res['n_accept']), n_call=c(res['n_call']), moved=c(res['moved']), scale=c(res['scale']))


def check_spline_fixture(out, g, o, xtol=1e-4, ltol=2e-4):
    """`out` (spline_fixture_launch) against the reference's trace `g`: every accept / reject decision (the counters), every
    state of every chain, the final scale.  A kernel sums in another order than torch, so ONE decision within rounding of its
    threshold may fall the other way; then -- and only then -- the walker that leaves the reference's chain first is held to
    assert_borderline on the oracle's margins (oracle object `o`), the chains are compared up to that step, and the counters may
    differ by what one flipped decision changes.  Returns True if every decision matched."""
    C, S = g['dz'].shape[1], g['dz'].shape[0]
    hx, hl = out['hist_x'][:C], out['hist_logl'][:C]
    rel = lambda a, b: float(np.max(np.abs(np.asarray(a, np.float64) - b) / (1.0 + np.abs(b))))
    assert rel(out['z0'][:C], g['latent'][:, 0]) < 3e-5
    same = int(out['n_call'][:C].sum()) == int(g['ncall']) and int(out['n_accept'][:C].sum()) == int(g['total_accepted'])
    first = first_divergence(hx, g['samples'], tol=20 * xtol)
    if same and np.all(first < 0):
        assert rel(hx, g['samples']) < xtol and rel(out['x'][:C], g['samples'][:, -1]) < xtol
        assert rel(hl, g['loglikes']) < ltol and rel(out['logl'][:C], g['loglikes'][:, -1]) < ltol
        assert rel(out['z'][:C], g['latent'][:, -1]) < xtol
        assert abs(float(out['scale'][0]) - float(g['scale_out'])) < 1e-5 * float(g['scale_out'])
        # the reference's usable-chain test on its own trace (nested.py:432)
        assert np.array_equal(out['moved'][:C], np.all(g['samples'][:, 0] != g['samples'][:, -1], axis=1))
        return True
    margins = np.empty((S, C))
    tr = spline_mcmc_trace(o, out['z0'][:C], g['init_logl'], float(g['loglstar']), float(g['step']), g['dz'], g['u'],
                           adapt=bool(g['dynamic']), like_scale=float(g['scale']), margins=margins)
    assert tr['ncall'] == int(g['ncall']) and tr['nacc'] == int(g['total_accepted'])     # the oracle IS the reference's chain
    s_first, w_first = min((int(s_), int(k)) for k, s_ in enumerate(first) if s_ >= 1)
    assert_borderline(hx, g['samples'], margins, [w_first])
    assert rel(hx[:, :s_first], g['samples'][:, :s_first]) < xtol                        # identical up to the flipped decision
    return False
