"""Developer diagnostic: one training call of two epochs through the NNEST_STAMP build of the rows form (tools/build_stamp.sh ->
tools/ab/lib_ROWS_STAMP.so); prints the kernels' own timelines of the last launches.   NNEST_HIP_LIB=tools/ab/lib_ROWS_STAMP.so python tools/stamp_rows.py [D]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnest_amd.spline import HipSpline
D = int(sys.argv[1]) if len(sys.argv) > 1 else 50
sp = HipSpline(D, 16, 3, seed=0)
rng = np.random.RandomState(0)
live = rng.uniform(-1, 1, size=(1000, D))
perms = torch.stack([torch.randperm(900) for _ in range(2)]).int()
sp.train_epochs(live[100:], live[:100], perms, None, seed=1, jitter=0.01, batch=100, max_epochs=2, patience=50)
torch.cuda.synchronize()
