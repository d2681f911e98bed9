"""K4 under the product step rule, repeated launches: reports every launch whose bounded waits ran out (error word of the sync buffer)\nand dumps the per-step counters around the failing step (developer diagnostic, round 4: python tools/stress_k4_sync.py explicit|warm4|fixed)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnest_amd import flow
D, C, S = 50, 1000, 250
nvp = flow.HipNVP(D, 16, 3, 1, seed=0)
u0 = np.random.RandomState(0).uniform(-1, 1, size=(C, D))
z0, _ = nvp.forward(u0)
l0 = flow.loglike(0, u0, 5.0)
star, step = float(l0.min()), 1 / np.sqrt(D)
which = sys.argv[1]
kw = dict(warm4=dict(form='solo', dynamic='batch', lag=8, warm=4), explicit=dict(form='solo', dynamic='batch', lag=8, warm=16), fixed=dict(form='solo'))[which]
for k in range(300):
    z, l = z0.clone(), l0.clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = nvp.mh_steps(0, 5.0, z, l, star, step, S, seed=k, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    v = int(res['sync'][-1].item()) & 0xffffffff if res['sync'] is not None else 0
    if v or dt > 0.01:
        sy = res['sync'].cpu().numpy().view(np.uint64)
        for it in range(14, 24):
            w = sy[(it * 8) * 8:(it * 8 + 8) * 8:8]
            tot = int(w.sum())
            print('   step', it, 'arrivals', tot >> 32, 'accepted', tot & 0xffffffff, 'per shard arrivals', [int(x) >> 32 for x in w])
        base = 2 * (S + 2) * 64
        print('   window 0 replicas:', [(int(sy[base + r * 8]) >> 32, hex(int(sy[base + r * 8]) & 0xffffffff)) for r in range(8)])
        print(which, k, 'code', v & 255, 'step', (v >> 8) & 4095, 'wg', v >> 20, 'ms', round(dt * 1e3, 1), flush=True)
print(which, 'done', flush=True)
