"""Developer diagnostic: wall-time split of a full run of a BASELINE configuration (low-overhead timers around the main segments).
   python tools/run_timing.py [flow] [config: 2 (default) | 3 | 4 | 5]"""
import os, sys, time, tempfile, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnest_amd.flow as nflow
import nnest_amd.trainer as tmod
import nnest_amd.sampler as smod
from nnest_amd.likelihoods import GaussianMix, Himmelblau, Rosenbrock
from nnest_amd.nested import NestedSampler
T = {}
def wrap(obj, name, key, sync=False):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        if sync:
            torch.cuda.synchronize()
        d = T.setdefault(key, [0.0, 0]); d[0] += time.perf_counter() - t0; d[1] += 1
        return r
    setattr(obj, name, g)
wrap(nflow, '_as_dev_f32', 'h2d(flow)')
_orig_h2d = tmod._as_dev_f32
def _h2d_split(x, device):
    device = torch.device(device)
    t0 = time.perf_counter(); torch.cuda.synchronize(); t1 = time.perf_counter()
    a = np.ascontiguousarray(x, dtype=np.float32); t2 = time.perf_counter()
    out = torch.empty(a.shape, dtype=torch.float32, device=device); t3 = time.perf_counter()
    stage = nflow._staging_f32(a.size, device); np.copyto(stage.numpy(), a.reshape(-1)); t4 = time.perf_counter()
    out.view(-1).copy_(stage, non_blocking=True); t5 = time.perf_counter()
    torch.cuda.current_stream(device).synchronize(); t6 = time.perf_counter()
    for key, d in (('h2d(trainer) pending GPU work before it', t1 - t0), ('h2d(trainer) host cast', t2 - t1), ('h2d(trainer) torch.empty', t3 - t2),
                   ('h2d(trainer) host copy to pinned', t4 - t3), ('h2d(trainer) copy_ call', t5 - t4), ('h2d(trainer) stream sync', t6 - t5)):
        e = T.setdefault(key, [0.0, 0]); e[0] += d; e[1] += 1
    return out
tmod._as_dev_f32 = _h2d_split
wrap(tmod.Trainer, '_training_jitter_launch', 'training_jitter (upload + launch)')
wrap(tmod.Trainer, 'train', 'Trainer.train')
_te = nflow.HipNVP.train_epochs
def _te_events(self, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = _te(self, *a, **k); e1.record(); torch.cuda.synchronize()
    d = T.setdefault('  train_epochs by GPU events', [0.0, 0]); d[0] += e0.elapsed_time(e1) * 1e-3; d[1] += 1
    return r
nflow.HipNVP.train_epochs = _te_events
wrap(nflow.HipNVP, 'train_epochs', 'train_epochs (K5 + readback)')
wrap(smod.Sampler, '_mcmc_sample', 'mcmc_sample')
wrap(smod.Sampler, '_rejection_prior_sample', 'rejection_prior_sample')
wrap(smod.Sampler, '_save_samples', 'save_samples')
wrap(np, 'save', 'np.save'); wrap(torch, 'save', 'torch.save')
# split train_epochs: everything up to the return of the launch call / the wait for the result
_orig_te = nflow.HipNVP.train_epochs.__wrapped__ if hasattr(nflow.HipNVP.train_epochs, '__wrapped__') else None
import nnest_amd._lib as L
_lib_check = L.check
def _check_timed(rc):
    T.setdefault('  K5 launch call returned at', [0.0, 0]); return _lib_check(rc)
_cpu = torch.Tensor.cpu
def _cpu_timed(self, *a, **k):
    t0 = time.perf_counter(); r = _cpu(self, *a, **k)
    if self.numel() == 6 and self.dtype == torch.int32:
        e = T.setdefault('  wait for the K5 result (.cpu)', [0.0, 0]); e[0] += time.perf_counter() - t0; e[1] += 1
    return r
torch.Tensor.cpu = _cpu_timed
_rand = torch.rand
def _rand_timed(*a, **k):
    t0 = time.perf_counter(); r = _rand(*a, **k); e = T.setdefault('  torch.rand (perm table)', [0.0, 0]); e[0] += time.perf_counter() - t0; e[1] += 1; return r
torch.rand = _rand_timed
_argsort = torch.Tensor.argsort
def _argsort_timed(self, *a, **k):
    t0 = time.perf_counter(); r = _argsort(self, *a, **k); e = T.setdefault('  argsort (perm table)', [0.0, 0]); e[0] += time.perf_counter() - t0; e[1] += 1; return r
torch.Tensor.argsort = _argsort_timed
flow = sys.argv[1] if len(sys.argv) > 1 else 'nvp'
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
like, scale, N = {2: (Rosenbrock(50), 5.0, 1000), 3: (GaussianMix(20), 10.0, 2000), 4: (Himmelblau(32), 5.0, 4000),
                  5: (Rosenbrock(100), 5.0, 8000)}[cfg]
_L = L.load()
_consume = _L.nnest_host_mcmc_consume
def _consume_timed(*a):
    t0 = time.perf_counter(); r = _consume(*a); e = T.setdefault('nnest_host_mcmc_consume (evidence loop)', [0.0, 0]); e[0] += time.perf_counter() - t0; e[1] += 1; return r
_L.nnest_host_mcmc_consume = _consume_timed
wrap(NestedSampler, '_mcmc_endpoints_fused', 'K4 batch (launch + read-back)')
wrap(NestedSampler, '_checkpoint', 'checkpoint')
if hasattr(NestedSampler, '_mcmc_loop_native'):
    wrap(NestedSampler, '_mcmc_loop_native', 'MCMC phase (native evidence loop + K4 + retrains)')
MARK = {}
_train0 = NestedSampler._train
def _train_marked(self, *a, **k):
    MARK.setdefault('first retrain', time.time())
    return _train0(self, *a, **k)
NestedSampler._train = _train_marked
np.random.seed(0); torch.manual_seed(0)
s = NestedSampler(like.x_dim, like, transform=lambda x: scale * x, log_dir=tempfile.mkdtemp(dir='/tmp'), num_live_points=N, log_level=30, flow=flow)
t0 = time.time()
s.run(mcmc_num_chains=N)
print('wall %.2f s logz %.3f' % (time.time() - t0, s.logz))
if 'first retrain' in MARK:
    print('  prior-rejection phase (run() up to the first retrain): %.2f s' % (MARK['first retrain'] - t0))
for k, (t, n) in sorted(T.items(), key=lambda kv: -kv[1][0]):
    print('  %-32s %8.2f s  %7d calls  %8.3f ms each' % (k, t, n, t / n * 1e3))
