"""Developer diagnostic: latency of the first GPU operation after the device has been idle for a while.
   python tools/idle_latency.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnest_amd.flow import _as_dev_f32
dev = torch.device('cuda', 0)
x = np.random.rand(900, 50)
_as_dev_f32(x, dev); torch.cuda.synchronize()
for idle_ms in (0, 1, 5, 20, 50, 200):
    ts = []
    for rep in range(8):
        time.sleep(idle_ms * 1e-3)
        t0 = time.perf_counter()
        y = _as_dev_f32(x, dev)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print('idle %4d ms: H2D of 900x50 f64 -> f32 takes %s ms' % (idle_ms, ' '.join('%.2f' % t for t in ts)))
# the same with a busy host instead of a sleeping one
for idle_ms in (20, 50):
    ts = []
    for rep in range(8):
        t1 = time.perf_counter()
        while (time.perf_counter() - t1) < idle_ms * 1e-3:
            pass
        t0 = time.perf_counter()
        y = _as_dev_f32(x, dev)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print('busy-wait %4d ms: %s ms' % (idle_ms, ' '.join('%.2f' % t for t in ts)))
