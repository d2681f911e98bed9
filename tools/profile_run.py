"""cProfile of a full nested-sampling run (developer diagnostic): where the host time goes.
  python tools/profile_run.py [x_dim] [flow] [chains]"""
import cProfile
import os
import pstats
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnest_amd.likelihoods import Rosenbrock  # noqa: E402
from nnest_amd.nested import NestedSampler  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 50
flow = sys.argv[2] if len(sys.argv) > 2 else 'nvp'
chains = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
np.random.seed(0)
torch.manual_seed(0)
tmp = tempfile.mkdtemp(dir='/tmp')
s = NestedSampler(D, Rosenbrock(D), transform=lambda x: 5 * x, log_dir=tmp, num_live_points=1000, log_level=30, flow=flow)
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
s.run(mcmc_num_chains=chains)
pr.disable()
print('wall %.1f s  logz %.3f +- %.3f  niter %d  ncall %d  retrains %d  batches %d' % (
    time.time() - t0, s.logz, s.logzerr, s.niter, s.ncall, s.num_retrains, s.num_batches))
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
