"""Developer diagnostic: BASELINE config 2 (Rosenbrock x_dim 50, 1000 live points) with the build-defined slice proposal
(NestedSampler(mcmc_proposal='slice')) over a few seeds -- log Z beside the Metropolis ensembles of tests/golden/logz_*_cfg2.json.
   python tools/run_slice_logz.py [n_seeds] [mcmc_steps]"""
import json, os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnest_amd.likelihoods import Rosenbrock, GaussianMix
from nnest_amd.nested import NestedSampler
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
cfg = int(sys.argv[3]) if len(sys.argv) > 3 else 2
prop = sys.argv[4] if len(sys.argv) > 4 else 'slice'
D, like, npts, scale = {1: (2, Rosenbrock(2), 100, 5.0), 2: (50, Rosenbrock(50), 1000, 5.0), 3: (20, GaussianMix(20), 2000, 10.0)}[cfg]
out = []
for seed in range(n):
    np.random.seed(seed); torch.manual_seed(seed)
    with tempfile.TemporaryDirectory() as tmp:
        s = NestedSampler(D, like, transform=lambda x: scale * x, log_dir=tmp, num_live_points=npts, log_level=30, flow='nvp',
                          mcmc_proposal=prop)
        t0 = time.time()
        s.run(mcmc_num_chains=npts, mcmc_steps=steps)
        out.append(dict(seed=seed, logz=float(s.logz), logzerr=float(s.logzerr), ncall=int(s.ncall), niter=int(s.niter), wall_s=time.time() - t0))
        print(out[-1], flush=True)
lz = np.array([o['logz'] for o in out])
print(json.dumps(dict(what='config %d with mcmc_proposal=%s, mcmc_steps=%d' % (cfg, prop, steps), mean=float(lz.mean()), std=float(lz.std(ddof=1)) if n > 1 else None,
                      stderr=float(lz.std(ddof=1) / np.sqrt(n)) if n > 1 else None, n=n, runs=out)))
