"""K5 (nnest_nvp_train, the one-row-per-wave kernel with its grid barriers and tagged weight publish), many launches over the
BASELINE shapes and ragged ones: every launch must come back without a bounded wait running out, and two launches from the same
state must produce the same bits (developer diagnostic, round 4).
   python tools/stress_k5_shapes.py [launches per shape, default 60]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnest_amd import flow
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for D, N, E in ((50, 1000, 30), (20, 2000, 20), (32, 4000, 8), (100, 8000, 4), (2, 100, 40), (5, 333, 30), (50, 777, 30), (64, 1234, 20), (100, 1000, 20)):
    rng = np.random.RandomState(N)
    live = rng.uniform(-1, 1, size=(N, D))
    nv = max(N // 10, 1)
    perms = torch.stack([torch.randperm(N - nv) for _ in range(E)]).int()
    ref, worst, t_all = None, 0.0, time.perf_counter()
    for k in range(n):
        nvp = flow.HipNVP(D, 16, 3, 1, seed=7)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        try:
            res = nvp.train_epochs(live[nv:], live[:nv], perms, None, max_epochs=E, seed=3, jitter=0.01, batch=100, patience=1000)
        except Exception as e:   # a wait ran out
            bad += 1
            print('   x_dim %d, %d live points, launch %d: %s' % (D, N, k, e), flush=True)
            continue
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        worst = max(worst, dt)
        w = np.asarray(nvp.store_packed())
        if ref is None:
            ref = w
        elif not np.array_equal(ref, w):
            bad += 1
            print('   x_dim %d, %d live points, launch %d: weights differ from the first launch (max %.3g)' % (D, N, k, np.abs(ref - w).max()), flush=True)
    print('x_dim %3d, %4d live points, %2d epochs: %d launches, mean %.3f ms per epoch, slowest launch %.3f ms per epoch' % (
        D, N, E, n, (time.perf_counter() - t_all) / n / E * 1e3, worst / E * 1e3), flush=True)
print('failures:', bad)
sys.exit(1 if bad else 0)
