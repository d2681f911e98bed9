"""MAF kernels: pass, proposal-kernel and training-epoch times (developer diagnostic; numbers quoted in DESIGN.md 3c).
  python tools/time_maf.py [x_dim] [n_live]
(under `rocprofv3 --kernel-trace --stats` the per-kernel split of a training minibatch comes out of the same run)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnest_amd import flow as nflow  # noqa: E402
from nnest_amd.maf import HipMAF  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 50
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
m = HipMAF(D, 16, 3, 1, seed=0)
rng = np.random.RandomState(0)
u = rng.uniform(-1, 1, size=(N, D))


def timed(fn, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts[1:]))


ud = nflow._as_dev_f32(u, m.device)
z, _ = m.forward(ud)
print('x_dim %d, %d rows, %d groups: forward %.1f us, inverse %.1f us' % (D, N, m.num_groups, timed(lambda: m.forward(ud)) * 1e3,
                                                                           timed(lambda: m.inverse(z)) * 1e3))
logl0 = nflow.loglike(0, u, 5.0)
S = 50
ms = timed(lambda: m.mh_steps(0, 5.0, z.clone(), logl0.clone(), float(logl0.min()), 1 / np.sqrt(D), S, seed=1), reps=4)
print('proposal kernel, %d walkers x %d steps: %.3f ms = %.2f us per step = %.3e evals/s' % (N, S, ms, ms * 1e3 / S, N * S / (ms * 1e-3)))
nv = N // 10
E = 8
perms = torch.stack([torch.randperm(N - nv) for _ in range(E)])
m.train_epochs(u[nv:], u[:nv], perms[:1], None, seed=1, jitter=0.01, batch=100, max_epochs=1, patience=50)
torch.cuda.synchronize()
t0 = time.perf_counter()
res = m.train_epochs(u[nv:], u[:nv], perms, None, seed=1, jitter=0.01, batch=100, max_epochs=E, patience=50)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / res['epochs_run']
nmb = (N - nv + 99) // 100
print('training: %.3f ms per epoch (%d minibatches: %.1f us each)' % (dt * 1e3, nmb, dt * 1e6 / nmb))
