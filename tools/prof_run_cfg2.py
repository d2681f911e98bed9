import cProfile, pstats, sys, os, tempfile, io
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from nnest_amd.likelihoods import Rosenbrock
from nnest_amd.nested import NestedSampler
like = Rosenbrock(50)
def run(seed):
    np.random.seed(seed); torch.manual_seed(seed)
    s = NestedSampler(50, like, transform=lambda x: 5.0 * x, log_dir=tempfile.mkdtemp(dir='/tmp'), num_live_points=1000, log_level=30, flow='nvp')
    s.run(mcmc_num_chains=1000)
    return s
run(1)
pr = cProfile.Profile(); pr.enable(); s = run(0); pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats('tottime').print_stats(38); print(st.getvalue()[:9000])
