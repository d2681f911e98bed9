"""Developer diagnostic: where the wall time of one Trainer.train call goes (NVP, config 2 sizes), with a busy host between calls.
   python tools/train_call_latency.py"""
import os, sys, time, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnest_amd.trainer import Trainer
import nnest_amd.flow as nflow
D = 50
tr = Trainer(D, hidden_dim=16, num_blocks=3, num_layers=1, flow='nvp', log_dir=tempfile.mkdtemp(dir='/tmp'), log_level=30)
rng = np.random.RandomState(0)
orig = nflow._as_dev_f32
acc = {'h2d': 0.0, 'n': 0}
def timed(x, device):
    t0 = time.perf_counter(); y = orig(x, device); torch.cuda.synchronize(); acc['h2d'] += time.perf_counter() - t0; acc['n'] += 1; return y
import nnest_amd.trainer as tmod
tmod._as_dev_f32 = timed
gaps = [0.0, 0.0, 0.03, 0.03, 0.03, 0.1, 0.1, 0.1, 0.3, 0.3, 0.0, 0.0]
for rep in range(len(gaps)):
    x = rng.uniform(-1, 1, size=(1000, D))
    t1 = time.perf_counter()
    while time.perf_counter() - t1 < gaps[rep]:
        pass
    acc['h2d'] = 0.0; acc['n'] = 0
    t0 = time.perf_counter()
    tr.train(x, max_iters=2000, jitter=-1 if hasattr(tr, 'training_jitter') else 0.01, patience=50)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('gap %.2f s, train call %d: %.1f ms wall, %d epochs (%.3f ms per epoch incl. everything); H2D copies %d x = %.2f ms' % (
        gaps[rep], rep, dt * 1e3, tr.losses.shape[0], dt * 1e3 / max(1, tr.losses.shape[0]), acc['n'], acc['h2d'] * 1e3))
