"""GPU parity for SingleSpeedNVP's scale variants (reference networks.py:289-347): scale='translate'
(translate-only couplings, logdet 0) and scale='constant' (translate-only couplings + a ScaleLayer scalar after each,
logdet += s): passes, fused proposal kernel, gradients and training against the golden fixtures produced by the
reference (tests/golden/scale_*.npz) and against the oracle.  Run with  pytest -m gpu."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')
from oracle import oracle as orc  # noqa: E402  (checker only)

G = os.path.join(os.path.dirname(__file__), 'golden')
FILES = sorted(glob.glob(os.path.join(G, 'scale_*.npz')))
IDS = [os.path.basename(p)[6:-4] for p in FILES]


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / (1.0 + np.abs(b))))


def cpu(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope='module')
def hip():
    from nnest_amd import flow
    if not torch.cuda.is_available():
        pytest.fail('no GPU visible: the -m gpu tests must run on an MI355X')
    return flow


def load(hip, path):
    g = np.load(path)
    scale = os.path.basename(path).split('_')[1]
    D, H, B, L = int(g['D']), int(g['H']), int(g['B']), int(g['L'])
    nvp = hip.HipNVP(D, H, B, L, scale=scale)
    return g, scale, nvp, orc.NVP(D, H, B, L, scale=scale)


@pytest.mark.parametrize('path', FILES, ids=IDS)
def test_passes_vs_reference_fixture(hip, path):
    g, scale, nvp, o = load(hip, path)
    assert list(nvp.state_dict().keys()) == [str(k) for k in g['keys']]     # the reference's state_dict keys, in order
    x = g['x']
    for tag in ('init', 'trained'):
        nvp.load_reference_vector(g['w_' + tag])
        assert np.array_equal(nvp.reference_vector(), g['w_' + tag])
        z, ldf = nvp.forward(x)
        assert rel(cpu(z), g['z_' + tag]) < 1e-5
        assert np.max(np.abs(cpu(ldf) - g['ldf_' + tag])) < 1e-6
        xb, ldi = nvp.inverse(g['z_' + tag])
        assert rel(cpu(xb), g['xb_' + tag]) < 1e-5
        assert np.max(np.abs(cpu(ldi) - g['ldi_' + tag])) < 1e-6
        assert rel(cpu(nvp.log_probs(x)), g['lp_' + tag]) < 2e-5
        xr, _ = nvp.inverse(z)
        assert float(torch.max(torch.abs(xr - torch.from_numpy(x.astype(np.float32)).cuda()))) <= 1e-5


@pytest.mark.parametrize('path', FILES, ids=IDS)
def test_gradients_and_adam_steps_vs_reference_fixture(hip, path):
    g, scale, nvp, o = load(hip, path)
    X, jitter = g['X'], float(g['jitter'])
    n = X.shape[0]
    nvp.load_reference_vector(g['w0'])
    data = X[g['perms'][0][:100]] + np.float32(jitter) * g['noises'][0][:100]
    loss, grad = nvp.loss_grad(data)
    assert abs(float(loss) - g['losses'][0]) < 3e-5 * (1 + abs(g['losses'][0]))
    gref = g['grads'][0]
    grad = cpu(grad)
    assert np.max(np.abs(nvp.reference_vector(grad) - gref)) < 1e-4 * (1e-3 + np.max(np.abs(gref)))
    unused = np.ones(nvp.num_params, bool)
    unused[orc.reference_index_map(nvp.D, nvp.H, nvp.B, nvp.L, scale)] = False
    assert np.all(grad[unused] == 0)
    # two epochs with the recorded shuffles and jitter noise (trainer.py:384-403)
    perms = torch.from_numpy(g['perms'].astype(np.int32))
    noises = torch.from_numpy(g['noises'])
    res = nvp.train_epochs(X, X[:23], perms, noises, jitter=jitter, batch=100, max_epochs=2, patience=50, finalize=False)
    losses = cpu(res['losses'])[:2, 0] * n
    ref = g['losses'].reshape(2, -1).sum(axis=1)
    np.testing.assert_allclose(losses, ref, rtol=3e-5)
    w = nvp.store_packed()
    assert np.all(w[unused] == 0)                      # the unused scale-net slots stay exactly zero under Adam
    dref = g['ws'][-1] - g['w0']
    dour = nvp.reference_vector(w) - g['w0']
    assert np.sqrt(np.mean((dour - dref) ** 2)) < 0.03 * np.sqrt(np.mean(dref ** 2))
    if scale == 'constant':
        k = orc.reference_index_map(nvp.D, nvp.H, nvp.B, nvp.L, scale)
        tail = np.isin(k, np.arange(nvp.num_params - nvp.B, nvp.num_params))
        assert np.max(np.abs(dour[tail] - dref[tail])) < 0.05 * np.max(np.abs(dref[tail]))


@pytest.mark.parametrize('scale', ['translate', 'constant'])
@pytest.mark.parametrize('C', [40, 5000])
def test_fused_proposal_kernel_vs_oracle(hip, scale, C):
    """K4 on a scale variant: team / register forms for 'translate' (zero scale-net fragments), image form for
    'constant'; C = 5000 takes the many-tile launch."""
    D = 20
    nvp = hip.HipNVP(D, 16, 3, 1, scale=scale, seed=3)
    w = nvp.store_packed() * 1.5
    if scale == 'constant':
        w[-3:] = [0.2, -0.15, 0.1]
    nvp.load_packed(w)
    o = orc.NVP(D, 16, 3, 1, nvp.store_packed(), scale=scale)
    rng = np.random.RandomState(C)
    S = 10
    init = rng.uniform(-0.6, 0.6, size=(C, D))
    init_logl = orc.loglike('rosenbrock', init, 5.0)
    dz, u = nvp.fill_noise(S, C, seed=5)
    z, _ = nvp.forward(init)
    logl = torch.from_numpy(init_logl).cuda()
    res = nvp.mh_steps(0, 5.0, z, logl, -1e12, 0.05, S, seed=5, history=True, dynamic='group')
    bad = 0
    groups = list(range(0, min(C, 160), 16))
    for g0 in groups:
        sl = slice(g0, min(g0 + 16, C))
        so, _, lo, sc, ncall, (acc, rej) = orc.mcmc_sample(o, 'rosenbrock', 5.0, init[sl], init_logl[sl], -1e12, 0.05, True,
                                                           cpu(dz)[:, sl], cpu(u)[:, sl])
        if int(res['n_call'][sl].sum()) == ncall and int(res['n_accept'][sl].sum()) == acc:
            assert rel(cpu(res['hist_x'])[sl], so) < 3e-4
            hl = cpu(res['hist_logl'])[sl]
            assert np.max(np.abs(hl - lo)) < 2e-3 * (1.0 + np.max(np.abs(lo)))
        else:
            bad += 1
    assert bad <= 1
    z2, _ = nvp.forward(init)
    logl2 = torch.from_numpy(init_logl).cuda()
    res2 = nvp.mh_steps(0, 5.0, z2, logl2, -1e12, 0.05, S, seed=5, dynamic='group')
    assert torch.equal(z2, z) and torch.equal(logl2, logl) and torch.equal(res2['x'], res['x'])


@pytest.mark.parametrize('scale', ['translate', 'constant'])
def test_nested_run_with_scale_variant(tmp_path, scale):
    import math
    from nnest_amd.likelihoods import Rosenbrock
    from nnest_amd.nested import NestedSampler
    np.random.seed(2)
    torch.manual_seed(2)
    s = NestedSampler(2, Rosenbrock(2), transform=lambda x: 5 * x, log_dir=str(tmp_path), num_live_points=400, log_level=30,
                      scale=scale, flow='nvp')
    assert s._fused_like_id is not None and s.trainer.netG.scale == scale
    s.run(mcmc_num_chains=40, train_iters=300)
    logz = math.log(math.pi / 10 * (1 - 0.5 * math.erfc(math.sqrt(5) - 1)) / 100)
    assert abs(s.logz - logz) <= 0.3, s.logz
