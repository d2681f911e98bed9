"""Timing of the spline-flow kernels on the GPU (developer diagnostic; numbers quoted in DESIGN.md).
  python tools/time_spline.py
K4-spline: constrained-MH launch, Rosenbrock x_dim=50, 1000 walkers x 250 steps (BASELINE config 2 with the reference's
default flow); also a chip-filling population.  Training: epochs at 1000 live points (9 minibatches + validation)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnest_amd import flow as nflow  # noqa: E402
from nnest_amd.spline import HipSpline  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    only_d = int(sys.argv[1]) if len(sys.argv) > 1 else None      # (python tools/time_spline.py [D] [walkers]: one case, K4 only)
    only_c = int(sys.argv[2]) if len(sys.argv) > 2 else None
    for D in (50, 20):
        if only_d is not None and D != only_d:
            continue
        sp = HipSpline(D, 16, 3, seed=0)
        rng = np.random.RandomState(0)
        live = rng.uniform(-1, 1, size=(1000, D))
        sp.actnorm_init(live[:100])
        for C, S in ((1000, 5 * D), (65536, 10)):
            if only_c is not None and C != only_c:
                continue
            u0 = rng.uniform(-1, 1, size=(C, D))
            z0, _ = sp.forward(u0)
            logl0 = nflow.loglike(0, u0, 5.0, device=dev)
            ts = []
            for rep in range(4):
                z, logl = z0.clone(), logl0.clone()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                res = sp.mh_steps(0, 5.0, z, logl, float(logl0.min()), 1 / np.sqrt(D), S, seed=rep)
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            ms = float(np.median(ts[1:]))
            print('K4-spline D=%d walkers=%d steps=%d: %.3f ms per launch -> %.3e evals/s (accept %.2f)' % (
                D, C, S, ms, C * S / (ms * 1e-3), float(res['n_accept'].sum()) / (C * S)))
        if only_c is not None:
            continue
        # training
        E = 40
        perms = torch.stack([torch.randperm(900) for _ in range(E)]).int()
        sp.train_epochs(live[100:], live[:100], perms[:2], None, seed=1, jitter=0.01, batch=100, max_epochs=2, patience=50)  # allocations
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = sp.train_epochs(live[100:], live[:100], perms, None, seed=1, jitter=0.01, batch=100, max_epochs=E, patience=50)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print('train D=%d N=1000: %.2f ms per epoch (9 minibatches + validation), loss %.3f -> %.3f' % (
            D, dt / E * 1e3, res['losses'][0, 0], res['losses'][res['epochs_run'] - 1, 0]))


if __name__ == '__main__':
    main()
