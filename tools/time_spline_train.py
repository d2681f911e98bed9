"""Developer diagnostic: repeated spline training calls in one process (per-call ms per epoch), to see the run-to-run spread.
   python tools/time_spline_train.py [D]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnest_amd.spline import HipSpline
D = int(sys.argv[1]) if len(sys.argv) > 1 else 50
sp = HipSpline(D, 16, 3, seed=0)
rng = np.random.RandomState(0)
live = rng.uniform(-1, 1, size=(1000, D))
E = 40
perms = torch.stack([torch.randperm(900) for _ in range(E)]).int()
sp.train_epochs(live[100:], live[:100], perms[:2], None, seed=1, jitter=0.01, batch=100, max_epochs=2, patience=50)
out = []
for rep in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = sp.train_epochs(live[100:], live[:100], perms, None, seed=rep, jitter=0.01, batch=100, max_epochs=E, patience=50)
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / res['epochs_run'] * 1e3)
print('D=%d ms per epoch over 8 calls:' % D, ' '.join('%.2f' % v for v in out))
