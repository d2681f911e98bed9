import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from nnest_amd import flow
for (D, N, batch, E) in [(5, 40, 100, 1), (5, 100, 100, 1), (50, 100, 100, 1), (50, 900, 100, 1), (50, 900, 100, 3)]:
    rng = np.random.RandomState(0)
    live = rng.normal(size=(N + 16, D)) * 0.3
    perms = torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(e)) for e in range(E)]).int()
    out = {}
    for one_cu in (True, False):
        nvp = flow.HipNVP(D, 16, 3, 1, seed=5)
        r = nvp.train_epochs(live[16:], live[:16], perms, None, max_epochs=E, seed=7, jitter=0.02, batch=batch, patience=100, one_cu=one_cu)
        m, v = nvp.adam_moments()
        out[one_cu] = (nvp.store_packed(), m, v, r['losses'].cpu().numpy())
    w0, w1 = out[True][0], out[False][0]
    m0, m1 = out[True][1], out[False][1]
    bad = np.flatnonzero(m0 != m1)
    ns = w0.size // 6
    print('D', D, 'N', N, 'E', E, 'weights differ:', int(np.sum(w0 != w1)), 'of', w0.size, ' exp_avg differ:', bad.size, ' max |dm|', float(np.max(np.abs(m0 - m1))) if bad.size else 0.0,
          ' losses', out[True][3][:E, 0], out[False][3][:E, 0])
    if bad.size:
        reg = bad // ns
        off = bad % ns
        H = 16
        names = []
        for o in off[:2000]:
            names.append('W0' if o < H * D else 'b0' if o < H * D + H else 'W1' if o < H * D + H + H * H else 'b1' if o < H * D + 2 * H + H * H else 'Wo' if o < H * D + 2 * H + H * H + D * H else 'bo')
        import collections
        print('   by (block,net):', collections.Counter(reg.tolist()), ' by layer:', collections.Counter(names))
