"""log Z under the three step-size rules (per-16-walker groups, batch-wide at lag 0 = the reference's rule, batch-wide at the
default lag) over a few seeds: a developer study of what the rule does to the evidence (GPU).
  python tools/step_rule_study.py [flow] [D] [N] [chains] [mcmc_steps] [nseeds]"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnest_amd.likelihoods import Rosenbrock  # noqa: E402
from nnest_amd.nested import NestedSampler  # noqa: E402

flow = sys.argv[1] if len(sys.argv) > 1 else 'spline'
D, N, C, S, K = [int(v) for v in (sys.argv[2:7] + ['10', '1000', '100', '50', '6'][len(sys.argv[2:7]):])]
out = {}
for mode in ('group', 'batch0', 'batch2', 'fixed'):
    zs, sc = [], []
    for seed in range(K):
        np.random.seed(seed)
        torch.manual_seed(seed)
        s = NestedSampler(D, Rosenbrock(D), transform=lambda x: 5.0 * x, log_dir=tempfile.mkdtemp(dir='/tmp'), num_live_points=N,
                          log_level=40, flow=flow)
        if mode == 'group':
            s._batch_rule_ok = False
        elif mode.startswith('batch'):
            s.mcmc_step_lag = int(mode[5:])
        s.run(mcmc_steps=S, mcmc_num_chains=C, mcmc_dynamic_step_size=mode != 'fixed')
        zs.append(float(s.logz))
        sc.append(float(s.total_accepted / max(1, s.total_accepted + s.total_rejected)))
    out[mode] = dict(mean=float(np.mean(zs)), sem=float(np.std(zs, ddof=1) / np.sqrt(K)), logz=zs, acceptance=float(np.mean(sc)))
    print(mode, json.dumps(out[mode]), flush=True)
