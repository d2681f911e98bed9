"""K4 at the BASELINE population under every step rule / kernel form (developer diagnostic).
  python tools/time_quad.py                                   launch times
  NNEST_HIP_LIB=tools/ab/lib_STAMP.so python tools/time_quad.py stamp     cycles per step segment of the quad form (NNEST_STAMP build)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnest_amd import flow  # noqa: E402

D, C, S = 50, int(os.environ.get('WALKERS', 1000)), 250
nvp = flow.HipNVP(D, 16, 3, 1, seed=0)
u0 = np.random.RandomState(0).uniform(-1, 1, size=(C, D))
z0, _ = nvp.forward(u0)
l0 = flow.loglike(0, u0, 5.0)
star, step = float(l0.min()), 1 / np.sqrt(D)


def timed(**kw):
    ts = []
    for k in range(5):
        z, l = z0.clone(), l0.clone()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        res = nvp.mh_steps(0, 5.0, z, l, star, step, S, seed=k, **kw)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
        nvp.check_sync(res)
    return float(np.median(ts[1:])), res


if len(sys.argv) > 1 and sys.argv[1] == 'stamp':
    for name, kw in (('solo fixed', dict(form='solo')), ('solo batch lag 4', dict(form='solo', warm=0, dynamic='batch', lag=4)),
                     ('solo batch lag 8', dict(form='solo', warm=0, dynamic='batch', lag=8)), ('solo product rule', dict(form='solo', dynamic='batch'))):
        ms, res = timed(**kw)
        o = res['scale'].cpu().numpy()
        print('%-18s %.3f ms; cycles per step: total %d  propose+noise %d  inverse %d  post %d | noise wave: draws %d of %d per iteration | '
              'inverse / post of the four net waves: %s' % (name, ms, o[0] / S, o[1] / S, o[2] / S, o[3] / S, o[4] / (S + 1), o[5] / (S + 1),
                                                           ' '.join('%d/%d' % (o[8 + 2 * j] / S, o[9 + 2 * j] / S) for j in range(4))))
    for name, kw in (('quad fixed', dict(form='quad')), ('quad1 fixed', dict(form='quad1')),
                     ('quad batch lag 2', dict(form='quad', dynamic='batch', lag=2)),
                     ('quad batch lag 4', dict(form='quad', dynamic='batch', lag=4)),
                     ('quad batch lag 0', dict(form='quad', dynamic='batch', lag=0))):
        ms, res = timed(**kw)
        o = res['scale'].cpu().numpy()
        print('%-18s %.3f ms; cycles per step: total %d  propose+noise %d  inverse %d  post %d  [tail %d | result wait %d | post atomic %d | '
              'scale update %d]' % (name, ms, o[0] / S, o[1] / S, o[2] / S, o[3] / S, o[4] / S, o[5] / S, o[6] / S, o[7] / S))
else:
    for name, kw in (('solo fixed', dict(form='solo')),
                     ('solo batch lag 3', dict(form='solo', warm=0, dynamic='batch', lag=3)), ('solo batch lag 4', dict(form='solo', warm=0, dynamic='batch', lag=4)),
                     ('solo batch lag 6', dict(form='solo', warm=0, dynamic='batch', lag=6)), ('solo batch lag 8', dict(form='solo', warm=0, dynamic='batch', lag=8)),
                     ('solo batch lag 12', dict(form='solo', warm=0, dynamic='batch', lag=12)),
                     ('solo batch lag 8 after 8 exact steps', dict(form='solo', warm=8, dynamic='batch', lag=8)),
                     ('solo batch lag 8 after 16 exact steps', dict(form='solo', warm=16, dynamic='batch', lag=8)),
                     ('solo batch lag 8 after 32 exact steps', dict(form='solo', warm=32, dynamic='batch', lag=8)),
                     ('solo batch lag 8 after 249 exact steps', dict(form='solo', warm=249, dynamic='batch', lag=8)),
                     ('default (as the sampler launches it)', dict(dynamic='batch')),
                     ('quad fixed', dict(form='quad')), ('quad1 fixed (both nets on one wave)', dict(form='quad1')),
                     ('quad1 batch lag 2', dict(form='quad1', dynamic='batch', lag=2)), ('quad batch lag 0', dict(form='quad', dynamic='batch', lag=0)),
                     ('quad batch lag 1', dict(form='quad', dynamic='batch', lag=1)),
                     ('quad batch lag 2', dict(form='quad', dynamic='batch', lag=2)),
                     ('quad batch lag 3', dict(form='quad', dynamic='batch', lag=3)),
                     ('quad batch lag 4', dict(form='quad', dynamic='batch', lag=4)),
                     ('quad batch lag 6', dict(form='quad', dynamic='batch', lag=6)),
                     ('team fixed', dict(form='team')), ('team group rule', dict(form='team', dynamic='group')),
                     ('team batch lag 2', dict(form='team', dynamic='batch', lag=2)),
                     ('team batch lag 4', dict(form='team', dynamic='batch', lag=4))):
        try:
            ms, _ = timed(**kw)
        except Exception as e:   # a form that does not take this population
            print('%-36s %s' % (name, str(e)[:60]))
            continue
        print('%-36s %.3f ms  %.2f us/step  %.3e evals/s' % (name, ms, ms * 1e3 / S, C * S / (ms * 1e-3)))
