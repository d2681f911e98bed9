"""K4 solo form: walkers (net waves) per workgroup 4 / 8 / 12 (developer diagnostic, round 4).
  python tools/time_solo_wpg.py [x_dim]        launch times per (WPG, population, step rule) + bitwise comparison of the chains
Each WPG runs in a child process (NNEST_SOLO_WPG is read once per process by the launcher)."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(D, wpg, out):
    import torch
    from nnest_amd import flow
    S = 250
    nvp = flow.HipNVP(D, 16, 3, 1, seed=0)
    res_out = {}
    rows = []
    for C in (wpg * 250, wpg * 125, 1000, 2000, 3000):
        if (C + wpg - 1) // wpg + 1 > 256:
            continue
        u0 = np.random.RandomState(0).uniform(-1, 1, size=(C, D))
        z0, _ = nvp.forward(u0)
        l0 = flow.loglike(0, u0, 5.0)
        star, step = float(l0.min()), 1 / np.sqrt(D)
        for name, kw in (('fixed', dict(form='solo')), ('lag8', dict(form='solo', dynamic='batch', lag=8, warm=0)),
                         ('product', dict(form='solo', dynamic='batch'))):
            ts = []
            for k in range(5):
                z, l = z0.clone(), l0.clone()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                res = nvp.mh_steps(0, 5.0, z, l, star, step, S, seed=3, **kw)
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
                nvp.check_sync(res)
            ms = float(np.median(ts[1:]))
            rows.append(dict(wpg=wpg, C=C, rule=name, ms=ms, us_per_step=ms * 1e3 / S, evals_per_s=C * S / (ms * 1e-3)))
            res_out['%d_%s_z' % (C, name)] = z.cpu().numpy()
            res_out['%d_%s_l' % (C, name)] = l.cpu().numpy()
            res_out['%d_%s_a' % (C, name)] = res['n_accept'].cpu().numpy()
            res_out['%d_%s_c' % (C, name)] = res['n_call'].cpu().numpy()
            res_out['%d_%s_s' % (C, name)] = res['scale'].cpu().numpy()
    np.savez(out, rows=json.dumps(rows), **res_out)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        child(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
        sys.exit(0)
    D = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    tmp = tempfile.mkdtemp()
    got = {}
    for wpg in (4, 8, 12):
        out = os.path.join(tmp, 'w%d.npz' % wpg)
        env = dict(os.environ, NNEST_SOLO_WPG=str(wpg))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), 'child', str(D), str(wpg), out], env=env, capture_output=True, text=True)
        if r.returncode != 0:
            print('WPG %d failed:\n%s' % (wpg, r.stderr[-2000:]))
            continue
        got[wpg] = np.load(out)
        for row in json.loads(str(got[wpg]['rows'])):
            print('x_dim %3d WPG %2d walkers %5d %-8s %.3f ms  %.2f us/step  %.3e evals/s' % (D, row['wpg'], row['C'], row['rule'], row['ms'],
                                                                                        row['us_per_step'], row['evals_per_s']))
    # the chains must not depend on how many walkers share a workgroup
    for wpg in (8, 12):
        if wpg not in got or 4 not in got:
            continue
        for k in got[4].files:
            if k == 'rows' or k not in got[wpg].files:
                continue
            same = np.array_equal(got[4][k], got[wpg][k], equal_nan=True)
            if not same:
                print('MISMATCH WPG %d vs 4: %s (max |diff| %g)' % (wpg, k, float(np.max(np.abs(got[4][k].astype(float) - got[wpg][k].astype(float))))))
        print('WPG %d vs 4: compared %d arrays' % (wpg, sum(1 for k in got[4].files if k != 'rows' and k in got[wpg].files)))
