"""[UNPINNED] The slice proposal in latent space (BASELINE.json north_star): ABSENT FROM THE REFERENCE
(nnest/sampler.py:310-316 proposes random-walk Metropolis moves only), so the step is build-defined (include/nnest_hip.h
nnest_slice_steps) and these tests hold the kernel to a CPU restatement of the same definition (oracle/oracle.py::slice_sample) on
the kernel's own directions and the shared Philox uniforms, to the invariants a slice-sampling update must keep, and to the
closed-form evidence of the reference's own test problem.  Run with  pytest -m gpu."""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')
from oracle import oracle as orc  # noqa: E402  (checker only)

G = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def hip():
    from nnest_amd import flow
    if not torch.cuda.is_available():
        pytest.fail('no GPU visible: the -m gpu tests must run on an MI355X')
    return flow


def cpu(t):
    return t.detach().cpu().numpy()


def setup(hip, D, C, seed=0):
    if D == 50:
        g = np.load(os.path.join(G, 'mcmc_rosen_d50.npz'))   # a trained flow of the reference (weights only)
        w = g['w']
    else:
        w = None
    nvp = hip.HipNVP(D, 16, 3, 1, seed=3)
    if w is not None:
        nvp.load_packed(w)
    w = nvp.store_packed()
    o = orc.NVP(D, 16, 3, 1, w)
    rng = np.random.RandomState(seed)
    u0 = rng.uniform(-0.6, 0.6, size=(C, D))
    l0 = orc.loglike('rosenbrock', u0, 5.0)
    star = float(np.quantile(l0, 0.2))
    keep = l0 > star
    u0, l0 = u0[keep], l0[keep]
    return nvp, o, u0, l0, star


@pytest.mark.parametrize('D,C,S', [(50, 48, 5), (2, 40, 8), (20, 33, 5), (100, 24, 3)])
def test_slice_kernel_vs_oracle_restatement(hip, D, C, S):
    """the kernel's chains against oracle.slice_sample on the kernel's own directions (nnest_slice_fill_noise) and the shared Philox
    uniforms: every walker whose counters agree must agree in every state; one that does not must have had a candidate within
    rounding of a decision threshold (box edge, slice level, L*) at the step where it leaves"""
    nvp, o, u0, l0, star = setup(hip, D, C)
    C = u0.shape[0]
    width = 2.0 / np.sqrt(D)
    seed, off = 4242, 100
    dz = nvp.fill_slice_noise(S, C, seed=seed, walker_offset=off)
    dzc = cpu(dz)
    assert abs(dzc.mean()) < 0.05 and abs(dzc.std() - 1) < 0.05
    z, _ = nvp.forward(u0)
    z0 = cpu(z).copy()
    logl = torch.from_numpy(l0).cuda()
    res = nvp.slice_steps(0, 5.0, z, logl, star, width, S, seed=seed, walker_offset=off, history=True)
    margins = np.empty((S, C))
    ref = orc.slice_sample(o, 'rosenbrock', 5.0, z0, l0, star, width, dzc, seed, walker_offset=off, margins=margins)
    hx = cpu(res['hist_x'])
    same = (cpu(res['n_eval']) == ref['n_eval']) & (cpu(res['n_call']) == ref['n_call']) & (cpu(res['n_move']) == ref['n_move'])
    assert same.mean() > 0.8, same.mean()
    err = np.max(np.abs(hx[same] - ref['x'][same]) / (1.0 + np.abs(ref['x'][same])))
    assert err < 2e-4, err
    assert np.max(np.abs(cpu(logl)[same] - ref['logl'][same]) / (1.0 + np.abs(ref['logl'][same]))) < 2e-4
    for c in np.flatnonzero(~same):   # a walker that took another decision: where it leaves, the oracle was within rounding of a threshold
        d = np.max(np.abs(hx[c] - ref['x'][c]) / (1.0 + np.abs(ref['x'][c])), axis=1) > 1e-3
        s_first = int(np.argmax(d)) if d.any() else S
        lo = max(s_first - 1, 0)
        assert np.min(margins[lo:min(s_first + 1, S), c]) < 2e-4, (c, s_first, margins[:, c])
    # the recorded-noise path replays the same launch
    z2 = torch.from_numpy(z0).cuda()
    logl2 = torch.from_numpy(l0).cuda()
    res2 = nvp.slice_steps(0, 5.0, z2, logl2, star, width, S, noise=dz, seed=seed, walker_offset=off)
    assert torch.equal(z2, z) and torch.equal(logl2, logl) and torch.equal(res2['n_eval'], res['n_eval'])


def test_slice_updates_keep_the_constraint_and_move(hip):
    """what a slice-sampling update under a hard constraint must do: every chain ends inside the box and above L*, nearly every update
    moves, the counters are consistent (evaluations >= counted calls >= moves), a repeated launch repeats its bits, a shard reproduces
    its slice of the full launch, and the end point's likelihood is the likelihood of the end point"""
    D, C, S = 50, 1000, 20
    nvp, o, u0, l0, star = setup(hip, D, C, seed=1)
    C = u0.shape[0]

    def run(lo, hi, off):
        z, _ = nvp.forward(u0[lo:hi])
        logl = torch.from_numpy(l0[lo:hi]).cuda()
        r = nvp.slice_steps(0, 5.0, z, logl, star, 2.0 / np.sqrt(D), S, seed=9, walker_offset=off)
        return cpu(z), cpu(logl), {k: cpu(v) for k, v in r.items() if v is not None}

    z, logl, r = run(0, C, 0)
    assert np.all(np.abs(r['x']) <= 1.0) and np.all(logl > star)
    assert np.all(r['n_eval'] >= r['n_call']) and np.all(r['n_call'] >= r['n_move']) and np.all(r['n_move'] <= S)
    assert r['n_move'].mean() > 0.95 * S                       # shrinkage ends on a point of the slice
    assert r['moved'].mean() > 0.99                            # the reference's usable-chain test (nested.py:432)
    assert 3.0 < r['n_eval'].mean() / S < 15.0                 # a handful of evaluations per update
    np.testing.assert_allclose(logl, orc.loglike('rosenbrock', r['x'], 5.0), rtol=2e-6, atol=1e-5)
    x_chk, _ = o.inverse(z)
    assert np.max(np.abs(x_chk - r['x'])) < 5e-5
    z2, logl2, r2 = run(0, C, 0)
    assert np.array_equal(z, z2) and np.array_equal(logl, logl2) and np.array_equal(r['n_eval'], r2['n_eval'])
    zs, ls, rs = run(256, 512, 256)
    assert np.array_equal(zs, z[256:512]) and np.array_equal(ls, logl[256:512])


def test_slice_shapes_outside_the_solo_layout_are_refused(hip):
    from nnest_amd import _lib
    for kw in (dict(num_hidden=32), dict(num_blocks=2), dict(num_layers=2), dict(scale='translate')):
        args = dict(num_inputs=6, num_hidden=16, num_blocks=3, num_layers=1)
        args.update(kw)
        nvp = hip.HipNVP(args['num_inputs'], args['num_hidden'], args['num_blocks'], args['num_layers'], seed=0, scale=args.get('scale', ''))
        z = torch.zeros(8, 6, device='cuda')
        logl = torch.zeros(8, dtype=torch.float64, device='cuda')
        with pytest.raises(_lib.NnestHipError):
            nvp.slice_steps(0, 5.0, z, logl, -1e9, 0.5, 2)


def test_nested_sampling_with_the_slice_proposal_rosenbrock_2d(tmp_path):
    """NestedSampler(mcmc_proposal='slice') on the reference's own integration problem (tests/test_nested.py:10-19: Rosenbrock 2-D,
    closed form log Z = -5.804, accepted within 0.2 there): the mean over seeds within 0.15, each run within 4 of its own error"""
    from nnest_amd.likelihoods import Rosenbrock
    from nnest_amd.nested import NestedSampler
    closed = math.log(math.pi / 10 * (1 - 0.5 * math.erfc(math.sqrt(5) - 1)) / 100)
    logz = []
    for seed in range(4):
        np.random.seed(seed)
        torch.manual_seed(seed)
        s = NestedSampler(2, Rosenbrock(2), transform=lambda x: 5.0 * x, log_dir=str(tmp_path / str(seed)), num_live_points=1000,
                          log_level=30, flow='nvp', mcmc_proposal='slice')
        assert s._fused_like_id is not None
        s.run(mcmc_num_chains=100, mcmc_steps=5, train_iters=500)
        assert abs(s.logz - closed) < 4 * s.logzerr + 0.05, (seed, s.logz, s.logzerr)
        logz.append(s.logz)
    assert abs(np.mean(logz) - closed) < 0.15, logz
