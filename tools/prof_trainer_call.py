"""Developer diagnostic: cProfile of repeated Trainer.train calls with the spline flow at config 2's shape (what a retrain costs
beside its epochs).   python tools/prof_trainer_call.py [n_calls]"""
import cProfile, os, pstats, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnest_amd.trainer import Trainer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
tmp = tempfile.mkdtemp(dir='/tmp')
t = Trainer(50, flow='spline', log_dir=tmp, log_level=30)
t.async_save = True
rng = np.random.RandomState(0)
live = rng.uniform(-1, 1, size=(1000, 50))
t.train(live, max_iters=2000, jitter=-1.0, patience=50)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter(); ep = 0
pr.enable()
for i in range(n):
    live = live + 0.001 * rng.standard_normal(live.shape)
    it0 = t.total_iters
    t.train(live, max_iters=2000, jitter=-1.0, patience=50)
    ep += t.total_iters - it0
pr.disable()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print('%d calls, %d epochs: %.2f ms per call, %.4f ms per epoch all in' % (n, ep, wall / n * 1e3, wall / ep * 1e3))
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
