"""K4 (NVP proposal kernel) launch time across populations (developer diagnostic): python tools/time_k4.py [x_dim] [form]
(NNEST_MH_OCC=1|2|3 in the environment pins the image form's build: waves per SIMD it is compiled for)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnest_amd import flow  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 50
FORM = sys.argv[2] if len(sys.argv) > 2 else None
nvp = flow.HipNVP(D, 16, 3, 1, seed=0)
rng = np.random.RandomState(0)
for C in (1000, 2000, 4000, 6000, 8000, 12000, 16000, 32768, 131072):
    S = int(os.environ.get('K4_STEPS', 50 if C <= 16000 else 20))
    u0 = rng.uniform(-1, 1, size=(C, D))
    z0, _ = nvp.forward(u0)
    l0 = flow.loglike(0, u0, 5.0)
    ts = []
    for k in range(4):
        z, l = z0.clone(), l0.clone()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        res = nvp.mh_steps(0, 5.0, z, l, float(l0.min()), 1 / np.sqrt(D), S, seed=k, form=FORM)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ms = float(np.median(ts[1:]))
    print('walkers %6d tiles %5d steps %3d: %.3f ms  %.2f us/step  %.3e evals/s  [%s]' % (C, (C + 15) // 16, S, ms, ms * 1e3 / S, C * S / (ms * 1e-3),
                                                                                      FORM or nvp.mh_form_for(C, False)))
