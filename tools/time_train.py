"""Developer timing of the training kernel (K5): epochs/s at the BASELINE configs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nnest_amd import flow
for D, N in ((2, 100), (50, 1000), (20, 2000), (100, 8000)):
    rng = np.random.RandomState(0)
    X = rng.uniform(-1, 1, size=(N, D)).astype(np.float32)
    nv = int(np.ceil(0.1 * N)); Xv, Xt = X[:nv], X[nv:]
    E = 40
    perm = torch.stack([torch.randperm(Xt.shape[0]) for _ in range(E)]).int().cuda()
    nvp = flow.HipNVP(D, 16, 3, 1, seed=0)
    xt, xv = torch.from_numpy(Xt).cuda(), torch.from_numpy(Xv).cuda()
    nvp.train_epochs(xt, xv, perm[:2], None, seed=1, jitter=0.01, max_epochs=2, patience=50)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = nvp.train_epochs(xt, xv, perm, None, seed=1, jitter=0.01, max_epochs=E, patience=1000)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    mb = (Xt.shape[0] + 99) // 100
    if os.environ.get('NNEST_HIP_LIB', '').endswith('STAMP.so'):
        ph = res['losses'].cpu().numpy().ravel()[:8] / (E * mb)
        print('   cycles/minibatch: fwd %d | bwd_s+stage %d | jobs_s %d | bwd_t %d | jobs_t %d | adam %d | rows+jitter (inside fwd) %d | validation(per epoch) %d' % (ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6], ph[7] * mb))
    print('D=%d N=%d: %.3f ms/epoch (%d minibatches, %.1f us/minibatch incl. validation share)' % (D, N, dt / E * 1e3, mb, dt / E / mb * 1e6))
