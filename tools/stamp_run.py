"""Developer diagnostic: per-segment cycle counts of the team MH kernel from an NNEST_STAMP build.
   NNEST_HIP_LIB=tools/ab/lib_STAMP.so python tools/stamp_run.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nnest_amd import flow
D, C, S = 50, 1000, 250
nvp = flow.HipNVP(D, 16, 3, 1, seed=0)
u0 = np.random.RandomState(0).uniform(-1, 1, size=(C, D))
z, _ = nvp.forward(u0)
logl = flow.loglike(0, u0, 5.0)
res = nvp.mh_steps(0, 5.0, z, logl, float(logl.min()), 1 / np.sqrt(D), S, seed=1)
o = res['scale'].cpu().numpy()
names = ['total', 'noise_wait', 'inverse', 'post', 'mlp', 'xch', 'update']
for r, tag in ((0, 'scale wave'), (8, 'translate wave')):
    print(tag, {n: round(float(o[r + i]) / S) for i, n in enumerate(names)}, 'cycles per step')
