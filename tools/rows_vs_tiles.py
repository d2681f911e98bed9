"""Developer diagnostic: the rows form of the spline training step against the tile form (NNEST_SPL_ROWS=0) in one run:
gradient difference, loss trajectory of a training call, ms per epoch.   python tools/rows_vs_tiles.py [D]"""
import os, subprocess, sys, json
import numpy as np
D = int(sys.argv[1]) if len(sys.argv) > 1 else 50
if len(sys.argv) > 2:
    import time, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from nnest_amd.spline import HipSpline
    sp = HipSpline(D, 16, 3, seed=0)
    rng = np.random.RandomState(0)
    live = rng.uniform(-1, 1, size=(1000, D))
    loss, grad = sp.loss_grad(live[:100])
    E = 40
    perms = torch.stack([torch.randperm(900, generator=torch.Generator().manual_seed(3)) for _ in range(E)]).int()
    res = sp.train_epochs(live[100:], live[:100], perms, None, seed=1, jitter=0.01, batch=100, max_epochs=E, patience=50)
    ts = []
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r2 = sp.train_epochs(live[100:], live[:100], perms, None, seed=rep, jitter=0.01, batch=100, max_epochs=E, patience=50)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / r2['epochs_run'] * 1e3)
    np.save(sys.argv[2], np.concatenate([[float(loss)], grad.cpu().numpy().astype(np.float64)]))
    L = res['losses'].numpy()
    print(json.dumps(dict(loss=float(loss), epochs=res['epochs_run'], train=[float(v) for v in L[:4, 0]] + [float(L[res['epochs_run'] - 1, 0])],
                          valid=[float(v) for v in L[:3, 1]] + [float(L[res['epochs_run'] - 1, 1])], ms_per_epoch=ts)))
    sys.exit(0)
out = {}
for tag, env in (('rows', {}), ('tiles', {'NNEST_SPL_ROWS': '0'})):
    e = dict(os.environ); e.update(env)
    f = '/tmp/rvt_%s.npy' % tag
    r = subprocess.run([sys.executable, __file__, str(D), f], env=e, capture_output=True, text=True)
    print(tag, r.stdout.strip()[-600:], r.stderr.strip()[-400:])
    out[tag] = np.load(f)
a, b = out['rows'], out['tiles']
d = np.abs(a[1:] - b[1:])
print('loss rows %.7f tiles %.7f | grad max |diff| %.3e at %d (|g| there %.3e), max |g| %.3e, identical: %s' % (a[0], b[0], d.max(), d.argmax(), abs(b[1:][d.argmax()]), np.abs(b[1:]).max(), np.array_equal(a, b)))
