"""Sweep every x_dim the kernels accept against the oracle (developer diagnostic, not part of the test suite):
forward / inverse / log_probs of both flows, the fused proposal kernel on a few steps."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnest_amd import flow  # noqa: E402
from nnest_amd.spline import HipSpline  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - b) / (1 + np.abs(b))))


worst = {}
rng = np.random.RandomState(0)
for D in range(1, 129):
    N = int(rng.choice([1, 7, 16, 17, 40, 100]))
    x = rng.uniform(-1, 1, size=(N, D)).astype(np.float32)
    nvp = flow.HipNVP(D, 16, 3, 1, seed=D)
    o = orc.NVP(D, 16, 3, 1, nvp.store_packed())
    z, ld = nvp.forward(x)
    z64, ld64 = o.forward(x, f64=True)
    xi, li = nvp.inverse(z64.astype(np.float32))
    xi64, li64 = o.inverse(z64.astype(np.float32), f64=True)
    e = max(rel(z.cpu().numpy(), z64), rel(ld.cpu().numpy(), ld64), rel(xi.cpu().numpy(), xi64), rel(li.cpu().numpy(), li64),
            rel(nvp.log_probs(x).cpu().numpy(), o.log_probs(x, f64=True)))
    worst['nvp'] = max(worst.get('nvp', (0, 0)), (e, D))
    if e > 2e-5:
        print('NVP D=%d N=%d err %.2e' % (D, N, e))
    # proposal kernel: same decisions as the oracle on a short chain
    if D >= 2:
        C = int(rng.choice([3, 16, 33]))
        init = rng.uniform(-0.5, 0.5, size=(C, D))
        l0 = orc.loglike('rosenbrock', init, 5.0)
        dz, u = nvp.fill_noise(4, C, seed=D)
        zz, _ = nvp.forward(init)
        ll = torch.from_numpy(l0).cuda()
        res = nvp.mh_steps(0, 5.0, zz, ll, -1e12, 0.05, 4, seed=D, history=True)
        bad = 0
        for g0 in range(0, C, 16):
            sl = slice(g0, min(g0 + 16, C))
            so, _, lo, _, ncall, (acc, rej) = orc.mcmc_sample(o, 'rosenbrock', 5.0, init[sl], l0[sl], -1e12, 0.05, False,
                                                              dz.cpu().numpy()[:, sl], u.cpu().numpy()[:, sl])
            if int(res['n_call'][sl].sum()) != ncall or int(res['n_accept'][sl].sum()) != acc:
                bad += 1
            elif rel(res['hist_x'].cpu().numpy()[sl], so) > 1e-4:
                print('NVP MH D=%d trace err %.2e' % (D, rel(res['hist_x'].cpu().numpy()[sl], so)))
        if bad > 1:
            print('NVP MH D=%d: %d groups with different decisions' % (D, bad))
    if D >= 2:
        sp = HipSpline(D, 16, 2, seed=D)
        sp.data_dep_init_done = True
        os_ = orc.Spline(D, 16, 2, 8, 3.0, sp.store_packed(), sp.P)
        z, ld = sp.forward(x)
        z64, ld64 = os_.forward(x, f64=True)
        z32, _ = os_.forward(x)
        xi, li = sp.inverse(z64.astype(np.float32))
        xi64, li64 = os_.inverse(z64.astype(np.float32), f64=True)
        xi32, _ = os_.inverse(z64.astype(np.float32))
        tol = max(3e-5, 4 * rel(z32, z64), 4 * rel(xi32, xi64))     # randn ActNorm scales amplify float32 rounding
        e = max(rel(z.cpu().numpy(), z64), rel(ld.cpu().numpy(), ld64), rel(xi.cpu().numpy(), xi64), rel(li.cpu().numpy(), li64))
        worst['spline'] = max(worst.get('spline', (0, 0)), (e / tol, D))
        if e > tol:
            print('SPLINE D=%d N=%d err %.2e (tol %.2e)' % (D, N, e, tol))
print('worst NVP error %.2e at D=%d; worst spline error / tolerance %.2f at D=%d' % (worst['nvp'] + worst['spline']))
