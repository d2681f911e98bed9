"""diagnostic: error of the HIP spline passes against the float64 oracle, next to the float32 oracle's own error"""
import glob, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from nnest_amd import spline


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - b) / (1 + np.abs(b))))


for path in sorted(glob.glob('tests/golden/spline_*.npz')):
    g = np.load(path)
    D, H, B, K = int(g['D']), int(g['H']), int(g['B']), int(g['K'])
    sp = spline.HipSpline(D, H, B, K, 3.0)
    for tag in ('init', 'trained'):
        sp.load_packed(g['w_' + tag], g['P'])
        sp.data_dep_init_done = True
        o = orc.Spline(D, H, B, K, 3.0, g['w_' + tag], g['P'])
        x = g['x']
        z64, ld64 = o.forward(x, f64=True)
        z32, ld32 = o.forward(x)
        z, ld = sp.forward(x)
        xi64, li64 = o.inverse(g['zs'], f64=True)
        xi32, li32 = o.inverse(g['zs'])
        xi, li = sp.inverse(g['zs'])
        xr, _ = sp.inverse(z)
        print('%-22s %-8s fwd z: hip %.1e orc32 %.1e ref %.1e | ld: hip %.1e orc32 %.1e | inv x: hip %.1e orc32 %.1e ref %.1e | ld hip %.1e | roundtrip hip %.1e' % (
            os.path.basename(path), tag, rel(z.cpu().numpy(), z64), rel(z32, z64), rel(g['z_' + tag], z64),
            rel(ld.cpu().numpy(), ld64), rel(ld32, ld64), rel(xi.cpu().numpy(), xi64), rel(xi32, xi64), rel(g['xs_' + tag], xi64),
            rel(li.cpu().numpy(), li64), rel(xr.cpu().numpy(), x.astype(np.float64))))
