"""Developer diagnostic: phase times of the spline gradient kernel from an NNEST_STAMP build of nnest_spline_train.hip.
   NNEST_HIP_LIB=tools/ab/libnnest_hip_STAMP.so python tools/stamp_spline.py [D]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nnest_amd.spline import HipSpline
D = int(sys.argv[1]) if len(sys.argv) > 1 else 50
sp = HipSpline(D, 16, 3, seed=0)
x = np.random.RandomState(0).uniform(-1, 1, size=(100, D))
sp.actnorm_init(x)
for rep in range(2):
    print('--- call', rep, flush=True)
    loss, grad = sp.loss_grad(x)
    torch.cuda.synchronize()
