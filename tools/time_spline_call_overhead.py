"""Developer diagnostic: the fixed cost of one spline training call (everything that is not an epoch): calls of 1, 2, 11 epochs.
   python tools/time_spline_call_overhead.py [D]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnest_amd.spline import HipSpline
D = int(sys.argv[1]) if len(sys.argv) > 1 else 50
sp = HipSpline(D, 16, 3, seed=0)
rng = np.random.RandomState(0)
live = rng.uniform(-1, 1, size=(1000, D))
perms = torch.stack([torch.randperm(900) for _ in range(16)]).int()
kw = dict(seed=1, jitter=0.01, batch=100, patience=50)
sp.train_epochs(live[100:], live[:100], perms[:2], None, max_epochs=2, **kw)
for E in (1, 2, 11):
    ts = []
    for rep in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sp.train_epochs(live[100:], live[:100], perms[:E], None, max_epochs=E, **kw)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print('max_epochs %2d: %.3f ms per call (median of 10)' % (E, float(np.median(ts))))
