import time, numpy as np, torch, sys
sys.path.insert(0,".")
from nnest_amd.spline import HipSpline
for D in (50,):
    sp = HipSpline(D, 16, 3, seed=0); rng=np.random.RandomState(0); live=rng.uniform(-1,1,size=(1000,D))
    perms = torch.stack([torch.randperm(900) for _ in range(40)]).int()
    kw=dict(seed=1, jitter=0.01, batch=100, patience=50)
    sp.train_epochs(live[100:], live[:100], perms[:2], None, max_epochs=2, **kw)
    ts=[]
    for r in range(12):
        torch.cuda.synchronize(); t0=time.perf_counter(); res=sp.train_epochs(live[100:], live[:100], perms, None, max_epochs=40, **kw); torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)/res["epochs_run"]*1e3)
    print("D=%d ms/epoch: median %.4f min %.4f" % (D, np.median(ts), min(ts)))
