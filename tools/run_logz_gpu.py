"""GPU-path log Z of a BASELINE configuration over a list of seeds -> gpurun_out/logz_gpu_cfg<cfg>.json (copied to
tests/golden/ and read by bench.py's `logz` report; the CPU-path counterpart is oracle/run_logz_cpu.py).
  python tools/run_logz_gpu.py 2 0,1,2,3,4,5 [lag] [tag]      lag: NNEST_MH_LAG of the batch-wide step rule ('-' or absent: the product's
                                                               default); tag: suffix of the output file; seeds may be a range a:b
  python tools/run_logz_gpu.py 2 0:48 - _maf maf              fifth argument: the flow ('nvp' | 'spline' | 'maf')"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nnest_amd import likelihoods  # noqa: E402
from nnest_amd.nested import NestedSampler  # noqa: E402

# 11 = config 1 at the reference's DEFAULT mcmc_num_chains = 10 (nested.py:185) instead of one chain per live point
CONFIGS = {1: ('Rosenbrock', 2, 5.0, 100, 100), 2: ('Rosenbrock', 50, 5.0, 1000, 1000), 3: ('GaussianMix', 20, 10.0, 2000, 2000),
           11: ('Rosenbrock', 2, 5.0, 100, 10)}
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
_sd = sys.argv[2] if len(sys.argv) > 2 else '0,1,2,3,4,5'
seeds = list(range(*[int(v) for v in _sd.split(':')])) if ':' in _sd else [int(v) for v in _sd.split(',')]
name, D, scale, N, chains = CONFIGS[cfg]
lag = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] != '-' else None
tag = sys.argv[4] if len(sys.argv) > 4 else ''
flow = sys.argv[5] if len(sys.argv) > 5 else 'nvp'


def build_stamp(sampler, chains, steps):
    """What produced the numbers (the GPU fixtures carried nothing that ties them to the kernels under test): the
    library's ABI version and the SHA-256 of the binary, and the defaults a run of this configuration takes -- K4 form, lag and
    exact warm-up steps of the step rule for `chains` walkers, the training kernel form.  tests/test_gpu_nested.py compares the
    defaults (not the hash) with the library under test."""
    import hashlib
    from nnest_amd import _lib
    netG = sampler.trainer.netG
    stamp = dict(abi_version=int(_lib.load().nnest_hip_version()), library_sha256=hashlib.sha256(open(_lib.LIB_PATH, 'rb').read()).hexdigest())
    if hasattr(netG, 'mh_form_for'):
        lag = netG.default_lag(chains) if steps >= 100 else 0
        stamp.update(mh_form=netG.mh_form_for(chains, dynamic='batch', lag=lag, warm=netG.default_warm(chains, 'batch', lag) if lag else 0),
                     step_lag=int(lag), step_warm=int(netG.default_warm(chains, 'batch', lag)) if lag else 0)
    stamp['train_form'] = os.environ.get('NNEST_TRAIN_FORM', 'rows')
    return stamp


runs = []
stamp = None
for seed in seeds:
    np.random.seed(seed)
    torch.manual_seed(seed)
    s = NestedSampler(D, getattr(likelihoods, name)(D), transform=lambda x: scale * x, log_dir=tempfile.mkdtemp(dir='/tmp'),
                      num_live_points=N, log_level=40, flow=flow)
    if stamp is None:
        stamp = build_stamp(s, chains, 5 * D)
        if lag is not None:
            stamp['step_lag'] = lag
            if lag == 0:
                stamp['step_warm'] = 0    # (every step exact: no warm-up in front of anything)
    t0 = time.time()
    s.run(mcmc_num_chains=chains, mcmc_step_lag=lag)
    runs.append(dict(seed=seed, logz=float(s.logz), logzerr=float(s.logzerr), h=float(s.h), niter=int(s.niter), ncall=int(s.ncall),
                     retrains=int(s.num_retrains), batches=int(s.num_batches), train_epochs_total=int(s.trainer.total_iters),
                     wall_s=time.time() - t0))
    print(json.dumps(runs[-1]), flush=True)
z = np.array([r['logz'] for r in runs])
doc = dict(what='GPU-path log Z: nnest_amd.NestedSampler on the HIP kernels (tools/run_logz_gpu.py)', config=cfg, likelihood=name, x_dim=D,
           num_live_points=N, mcmc_num_chains=chains, flow=flow + ' h16 b3 l1', train_iters=500, step_rule='batch-wide, default lag' if lag is None else 'batch-wide, lag %d' % lag,
           seeds=seeds, logz=z.tolist(), mean=float(z.mean()), std=float(z.std(ddof=1)) if len(z) > 1 else None,
           stderr=float(z.std(ddof=1) / np.sqrt(len(z))) if len(z) > 1 else None, build=stamp, runs=runs)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
with open(os.path.join(ROOT, 'gpurun_out', 'logz_gpu_cfg%d%s%s.json' % (cfg, '' if lag is None else '_lag%d' % lag, tag)), 'w') as f:
    json.dump(doc, f, indent=1)
print('mean %.3f  std %.3f  stderr %.3f' % (doc['mean'], doc['std'] or 0, doc['stderr'] or 0))
