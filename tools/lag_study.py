"""Developer diagnostic: what the lag of the batch-wide step rule does to ONE proposal batch of a config-2 run.
A config-2 run is taken to `iters` iterations, the state of its last MCMC batch is captured (flow, start points, threshold) and
the same batch is re-run under each rule: accepted moves per chain, squared displacement start -> end (unit-cube coordinates),
chains with every coordinate moved (what nested.py:432 calls usable), final scale.
   python tools/lag_study.py [iters] [reps]"""
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnest_amd.nested import NestedSampler  # noqa: E402
from nnest_amd.likelihoods import Rosenbrock  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
D, N = 50, 1000
np.random.seed(0)
torch.manual_seed(0)
s = NestedSampler(D, Rosenbrock(D), transform=lambda x: 5 * x, log_dir=tempfile.mkdtemp(), num_live_points=N, log_level=40, flow='nvp')
last = {}
orig = s._mcmc_endpoints_fused


def spy(mcmc_steps, step_size, dynamic, **kw):
    last.update(kw, mcmc_steps=mcmc_steps, step_size=step_size)
    return orig(mcmc_steps, step_size, dynamic, **kw)


s._mcmc_endpoints_fused = spy
s.run(strategy=['rejection_prior', 'mcmc'], mcmc_num_chains=N, max_iters=iters, train_iters=500)
netG = s.trainer.netG
u0, l0, star = last['init_samples'], np.asarray(last['init_loglikes'], dtype=np.float64), float(last['loglstar'])
print('state after %d iterations: loglstar %.3f, %d chains, step_size %.4f' % (s.niter, star, u0.shape[0], last['step_size']))
rows = [('fixed step', dict(dynamic=False), 250)]
for lag in (0, 3, 4, 8, 12, 15):
    rows.append(('batch rule, lag %d' % lag, dict(dynamic='batch', lag=lag, warm=0), 250))
for lag, warms in ((8, (4, 8, 16, 24, 32, 48, 64, 96, 128)), (6, (16, 32)), (4, (8, 16, 32)), (12, (32, 64))):
    for warm in warms:
        rows.append(('batch rule, lag %d, first %d steps exact' % (lag, warm), dict(dynamic='batch', lag=lag, warm=warm), 250))
for name, kw, S in rows:
    acc, disp, usable, scales, forms = [], [], [], [], set()
    for r in range(reps):
        z, _ = netG.forward(u0)
        logl = torch.as_tensor(l0).to(netG.device)
        res = netG.mh_steps(s._fused_like_id, s._linear_scale, z, logl, star, float(last['step_size']), S, seed=1000 + r, **kw)
        if kw.get('dynamic'):
            netG.check_sync(res)
        x = res['x'].double().cpu().numpy()
        acc.append(float(res['n_accept'].double().mean()))
        disp.append(float(np.mean(np.sum((x - u0) ** 2, axis=1))))
        usable.append(float(np.mean(np.all(x != u0, axis=1))))
        scales.append(float(res['scale'].double().mean()))
    se = lambda v: np.std(v, ddof=1) / np.sqrt(len(v))  # noqa: E731
    print('%-40s accepted/chain %7.2f +- %.2f   |dx|^2 %.5f +- %.5f   usable %.4f   final scale %.5f' % (
        name, np.mean(acc), se(acc), np.mean(disp), se(disp), np.mean(usable), np.mean(scales)))
