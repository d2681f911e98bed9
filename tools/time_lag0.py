"""Developer diagnostic: the reference's exact step rule (lag 0) on the solo form, 1000 walkers x 250 steps at config 2 --
kernel ms by HIP events; NNEST_SOLO_DUO=0 in the environment keeps round 5's two evaluations on one wave.
   python tools/time_lag0.py [x_dim] [walkers]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnest_amd import flow
D = int(sys.argv[1]) if len(sys.argv) > 1 else 50
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
S = 5 * D
nvp = flow.HipNVP(D, 16, 3, 1, seed=0)
u0 = np.random.RandomState(0).uniform(-1, 1, size=(C, D))
z0, _ = nvp.forward(u0)
l0 = flow.loglike(0, u0, 5.0)
star, step = float(l0.min()), 1 / np.sqrt(D)
for name, kw in (('lag 0 (exact rule)', dict(dynamic='batch', lag=0)), ('product rule', dict(dynamic='batch')), ('fixed step', dict())):
    ts, chk = [], None
    for k in range(60):
        z, l = z0.clone(), l0.clone()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        res = nvp.mh_steps(0, 5.0, z, l, star, step, S, seed=k % 7, **kw)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
        if k == 0:
            chk = (float(z.double().sum()), int(res['n_accept'].sum()), float(res['scale'][0]))
    flow.HipNVP.check_sync(res)
    print('%-20s %s: %.4f ms per %d x %d launch (median of the last 40), %.3f us/step; seed-0 chain checksum %r' % (
        name, nvp.mh_form_for(C, **{k_: v for k_, v in kw.items()}), float(np.median(ts[20:])), C, S, float(np.median(ts[20:])) * 1e3 / S, chk))
