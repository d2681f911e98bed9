"""K4 under the product's default step rule, many launches over the populations and widths the solo form serves (one, two and three
walker quartets per SIMD set; x_dim 20 / 32 / 50 / 100): every launch must come back without a bounded wait running out and within
a sane time (developer diagnostic, round 4: an intermittent hang hides from a test that launches a shape a handful of times).
   python tools/stress_k4_shapes.py [launches per shape, default 400]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnest_amd import flow
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
bad = 0
for D, C in ((50, 1000), (50, 2000), (50, 3000), (50, 777), (50, 1500), (20, 2000), (20, 1000), (32, 1000), (32, 3000), (100, 1000), (100, 2000)):
    S = 5 * D
    nvp = flow.HipNVP(D, 16, 3, 1, seed=D + C)
    u0 = np.random.RandomState(C).uniform(-1, 1, size=(C, D))
    z0, _ = nvp.forward(u0)
    l0 = flow.loglike(0, u0, 5.0)
    star, step = float(np.median(l0.cpu().numpy())), 1 / np.sqrt(D)
    lag = nvp.default_lag(C)
    form = nvp.mh_form_for(C, dynamic='batch', lag=lag, warm=nvp.default_warm(C, 'batch', lag))
    worst, t_all = 0.0, time.perf_counter()
    for k in range(n):
        z, l = z0.clone(), l0.clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = nvp.mh_steps(0, 5.0, z, l, star, step, S, seed=1000 * C + k, dynamic='batch')
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        worst = max(worst, dt)
        v = int(res['sync'][-1].item()) & 0xffffffff if res.get('sync') is not None else 0
        if v or dt > 0.02:
            bad += 1
            print('   x_dim %d, %d walkers, launch %d: code %d step %d wg %d, %.1f ms' % (D, C, k, v & 255, (v >> 8) & 4095, v >> 20, dt * 1e3), flush=True)
    print('x_dim %3d, %4d walkers, %3d steps [%s, lag %d]: %d launches, mean %.3f ms, slowest %.3f ms' % (
        D, C, S, form, lag, n, (time.perf_counter() - t_all) / n * 1e3, worst * 1e3), flush=True)
print('failures:', bad)
sys.exit(1 if bad else 0)
