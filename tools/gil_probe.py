"""Developer diagnostic: how much of a config-2 run's wall time is the main thread waiting for the interpreter lock while the
worker thread (utils.BackgroundJobs: scalars.csv rows, models/netG.pt) runs Python code.
   python tools/gil_probe.py
Runs the same seed (a) as shipped, (b) with a 0.1 ms switch interval, (c) with the worker's jobs dropped."""
import os, sys, time, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnest_amd.likelihoods import Rosenbrock
from nnest_amd.nested import NestedSampler
import nnest_amd.utils as U

like = Rosenbrock(50)


def run(seed=0):
    np.random.seed(seed); torch.manual_seed(seed)
    s = NestedSampler(50, like, transform=lambda x: 5.0 * x, log_dir=tempfile.mkdtemp(dir='/tmp'), num_live_points=1000, log_level=30, flow='nvp')
    t0 = time.time()
    s.run(mcmc_num_chains=1000)
    return time.time() - t0, s.logz


run(1)
for rep in range(2):
    print('as shipped                 wall %.3f s  logz %.3f' % run())
    old = sys.getswitchinterval()
    sys.setswitchinterval(1e-4)
    print('switch interval 0.1 ms     wall %.3f s  logz %.3f' % run())
    sys.setswitchinterval(old)
    submit = U.BackgroundJobs.submit
    U.BackgroundJobs.submit = lambda self, fn: None
    print('worker jobs dropped        wall %.3f s  logz %.3f' % run())
    U.BackgroundJobs.submit = submit
