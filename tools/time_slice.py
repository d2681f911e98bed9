"""Developer diagnostic: the slice proposal kernel (build-defined; nnest_slice_steps) at config 2's population -- ms per launch, flow
evaluations per second, evaluations per update, beside the Metropolis kernel's launch on the same walkers.
   python tools/time_slice.py [x_dim] [walkers] [updates]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnest_amd import flow
D = int(sys.argv[1]) if len(sys.argv) > 1 else 50
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
S = int(sys.argv[3]) if len(sys.argv) > 3 else 25
nvp = flow.HipNVP(D, 16, 3, 1, seed=0)
u0 = np.random.RandomState(0).uniform(-1, 1, size=(C, D))
z0, _ = nvp.forward(u0)
l0 = flow.loglike(0, u0, 5.0)
star, step = float(l0.min()), 1 / np.sqrt(D)
for width in (0.5 * step, step, 2 * step, 4 * step):
    ts = []
    for k in range(12):
        z, l = z0.clone(), l0.clone()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = nvp.slice_steps(0, 5.0, z, l, star, width, S, seed=k)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ms, ne = float(np.median(ts[4:])), int(r['n_eval'].sum())
    print('slice  width %.3f: %.3f ms per %d x %d launch, %.2f evals per update, %.3e evals/s, moved %.3f, n_call/n_eval %.2f' % (
        width, ms, C, S, ne / (C * S), ne / (ms * 1e-3), float(r['n_move'].sum()) / (C * S), float(r['n_call'].sum()) / ne))
ts = []
for k in range(12):
    z, l = z0.clone(), l0.clone()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    r = nvp.mh_steps(0, 5.0, z, l, star, step, 5 * D, seed=k)
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ms = float(np.median(ts[4:]))
print('metropolis (fixed step): %.3f ms per %d x %d launch, %.3e evals/s, accepted %.3f' % (ms, C, 5 * D, C * 5 * D / (ms * 1e-3), float(r['n_accept'].sum()) / (C * 5 * D)))
