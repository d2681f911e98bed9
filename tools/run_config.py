"""Run one BASELINE.json configuration end to end on this GPU and print a JSON summary (developer / evidence tool).
  python tools/run_config.py 3 [flow] [seed]      configs: 2 Rosenbrock-50/1000, 3 GaussianMix-20/2000, 4 Himmelblau-32/4000, 5 Rosenbrock-100/8000"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnest_amd.likelihoods import GaussianMix, Himmelblau, Rosenbrock  # noqa: E402
from nnest_amd.nested import NestedSampler  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
flow = sys.argv[2] if len(sys.argv) > 2 else 'nvp'
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
like, scale, N = {2: (Rosenbrock(50), 5.0, 1000), 3: (GaussianMix(20), 10.0, 2000), 4: (Himmelblau(32), 5.0, 4000),
                  5: (Rosenbrock(100), 5.0, 8000)}[cfg]
np.random.seed(seed)
torch.manual_seed(seed)
s = NestedSampler(like.x_dim, like, transform=lambda x: scale * x, log_dir=tempfile.mkdtemp(dir='/tmp'), num_live_points=N,
                  log_level=30, flow=flow)
t0 = time.time()
s.run(mcmc_num_chains=N)
out = dict(config=cfg, flow=flow, seed=seed, likelihood=type(like).__name__, x_dim=like.x_dim, num_live_points=N, mcmc_num_chains=N,
           wall_s=time.time() - t0, logz=s.logz, logzerr=s.logzerr, h=s.h, niter=s.niter, ncall=s.ncall, retrains=s.num_retrains,
           batches=s.num_batches, train_epochs_total=int(s.trainer.total_iters))
# the reference stops at max_iters = 1e6 whatever the state (nested.py:100): a run that reaches the cap has NOT converged -- the remaining
# live-point mass still dominates -- and what it carries as logz is not an evidence (config 5: it scatters by hundreds between
# seeds).  Such a run is reported as what it is: unconverged, with its cost per iteration (round-4 verdict).
out['converged'] = bool(s.niter < 1000000)
out['us_per_iteration'] = out['wall_s'] / max(1, s.niter) * 1e6
if not out['converged']:
    out['logz_at_the_iteration_cap_not_an_evidence'] = out.pop('logz')
    out.pop('logzerr')
if cfg == 3:
    out['logz_analytic'] = -20 * float(np.log(20.0))   # unit-mass mixture inside [-10,10]^20
print(json.dumps(out))
