"""Developer diagnostic: phase times of the spline training kernels from an NNEST_STAMP build (two epochs of two minibatches).
   NNEST_HIP_LIB=$PWD/tools/ab/lib_SPLSTAMP.so python tools/stamp_spline_train.py [D]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nnest_amd.spline import HipSpline
D = int(sys.argv[1]) if len(sys.argv) > 1 else 50
sp = HipSpline(D, 16, 3, seed=0)
live = np.random.RandomState(0).uniform(-1, 1, size=(300, D))
perms = torch.stack([torch.randperm(200) for _ in range(2)]).int()
sp.train_epochs(live[100:], live[:100], perms, None, seed=1, jitter=0.01, batch=100, max_epochs=2, patience=50)
torch.cuda.synchronize()
