"""GPU parity for the fast/slow hierarchy (reference FastSlowNVP, networks.py:86-150, :350-380; Trainer(num_slow=...)):
passes, every gradient element and Adam steps against fixtures produced by the reference (tests/golden/fastslow_*.npz),
the reference's own test (tests/test_flows.py:94-118) and a nested-sampling run.  Run with  pytest -m gpu."""
import glob
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

G = os.path.join(os.path.dirname(__file__), 'golden')
FILES = sorted(glob.glob(os.path.join(G, 'fastslow_*.npz')))
IDS = [os.path.basename(p)[9:-4] for p in FILES]


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / (1.0 + np.abs(b))))


def cpu(t):
    return t.detach().cpu().numpy()


def build(g):
    from nnest_amd.fastslow import HipFastSlowNVP
    if not torch.cuda.is_available():
        pytest.fail('no GPU visible: the -m gpu tests must run on an MI355X')
    return HipFastSlowNVP(int(g['F']), int(g['S']), 16, 3, 1)


@pytest.mark.parametrize('path', FILES, ids=IDS)
def test_passes_vs_reference_fixture(path):
    g = np.load(path)
    net = build(g)
    S = int(g['S'])
    assert list(net.state_dict().keys()) == [str(k) for k in g['keys']]
    x = g['x']
    for tag in ('init', 'trained'):
        net.load_packed(g['w_' + tag])
        assert np.array_equal(net.store_packed(), g['w_' + tag])          # round trip incl. the entries the masks never use
        z, ld = net.forward(x)
        assert rel(cpu(z), g['z_' + tag]) < 2e-5 and rel(cpu(ld), g['ldf_' + tag]) < 2e-5
        xb, ldi = net.inverse(g['z_' + tag])
        assert rel(cpu(xb), g['xb_' + tag]) < 2e-5 and rel(cpu(ldi), g['ldi_' + tag]) < 2e-5
        assert rel(cpu(net.log_probs(x)), g['lp_' + tag]) < 3e-5
        # a move of the fast latent block alone leaves the slow coordinates bit-identical (tests/test_flows.py:107-113)
        dz = torch.randn_like(z) * 0.01
        dz[:, :S] = 0.0
        x0, _ = net.inverse(z)
        xp, _ = net.inverse(z + dz)
        assert torch.equal(x0[:, :S], xp[:, :S]) and not torch.equal(x0[:, S:], xp[:, S:])


@pytest.mark.parametrize('path', FILES, ids=IDS)
def test_gradient_and_adam_steps_vs_reference(path):
    g = np.load(path)
    net = build(g)
    net.load_packed(g['w_init'])
    X, jitter = g['X'], float(g['jitter'])
    used = net.used_mask()
    data = X[g['perms'][0][:100]] + np.float32(jitter) * g['noises'][0][:100]
    loss, grads = net.loss_grad(data)
    assert abs(float(loss) - g['losses'][0]) < 3e-5 * (1 + abs(g['losses'][0]))
    gref = g['grads'][0]
    gour = net.reference_gradient(grads)
    assert np.max(np.abs(gour - gref)[used]) < 1e-4 * (1e-3 + np.max(np.abs(gref)))
    assert np.all(gref[~used] == 0) and np.all(gour[~used] == 0)
    res = net.train_epochs(X, X[:23], torch.from_numpy(g['perms'].astype(np.int64)), torch.from_numpy(g['noises']), jitter=jitter,
                           batch=100, max_epochs=2, patience=50)
    losses = res['losses'].numpy()[:2, 0] * X.shape[0]
    np.testing.assert_allclose(losses, g['losses'].reshape(2, -1).sum(axis=1), rtol=5e-5)
    if res['best_epoch'] == 2:
        dref = (g['ws'][-1] - g['w_init'])[used]
        dour = (net.store_packed() - g['w_init'])[used]
        assert np.sqrt(np.mean((dour - dref) ** 2)) < 0.05 * np.sqrt(np.mean(dref ** 2))


def test_reference_test_nvp_slow():
    """tests/test_flows.py:94-118 restated on this build's Trainer"""
    from nnest_amd.trainer import Trainer
    for num_slow in [2, 3, 4, 5]:
        for num_fast in [2, 5]:
            dims = num_slow + num_fast
            t = Trainer(dims, num_slow=num_slow, flow='nvp', log_dir=None, log_level=30)
            test_data = torch.from_numpy(np.random.normal(size=(10, dims))).float()
            z, z_log_det = t.forward(test_data)
            assert z.shape == torch.Size([10, dims]) and z_log_det.shape == torch.Size([10])
            x, x_log_det = t.inverse(z)
            assert abs(float(torch.max(x.cpu() - test_data))) <= 1e-5
            assert abs(float(torch.max(x_log_det + z_log_det))) <= 1e-5
            dz = torch.randn_like(z) * 0.01
            dz[:, 0:num_slow] = 0.0
            xp, _ = t.inverse(z + dz)
            assert float(torch.max((x - xp)[:, :num_slow].abs())) == 0
            assert t.get_synthetic_samples(10).shape == torch.Size([10, dims])
            assert t.log_probs(test_data).shape == torch.Size([10])


def test_nested_run_with_slow_block(tmp_path):
    from nnest_amd.likelihoods import Rosenbrock
    from nnest_amd.nested import NestedSampler
    np.random.seed(1)
    torch.manual_seed(1)
    s = NestedSampler(2, Rosenbrock(2), transform=lambda x: 5 * x, log_dir=str(tmp_path), num_live_points=200, log_level=30,
                      flow='nvp', num_slow=1)
    assert type(s.trainer.netG).__name__ == 'HipFastSlowNVP' and s._fused_like_id is None
    s.run(mcmc_num_chains=20, train_iters=100, mcmc_steps=10)
    assert abs(s.logz - math.log(math.pi / 10 * (1 - 0.5 * math.erfc(math.sqrt(5) - 1)) / 100)) <= 0.5, s.logz
    assert 0 < s.total_fast_calls < s.total_calls


# ---- FastSlowSpline (networks.py:718-731) -------------------------------------------------------------------------------
FSS_FILES = sorted(glob.glob(os.path.join(G, 'fastslowspline_*.npz')))


def build_spline(g):
    from nnest_amd.fastslow import HipFastSlowSpline
    net = HipFastSlowSpline(int(g['F']), int(g['S']), 16, 3)
    return net, {'fast': g['P_fast'], 'slow': g['P_slow']}


@pytest.mark.parametrize('path', FSS_FILES, ids=[os.path.basename(p)[15:-4] for p in FSS_FILES])
def test_fast_slow_spline_vs_reference_fixture(path):
    g = np.load(path)
    net, P = build_spline(g)
    S = int(g['S'])
    assert list(net.state_dict().keys()) == [str(k) for k in g['keys']]
    # the first forward batch initialises the ActNorms of both stages (networks.py:698-705)
    net.load_packed(g['w_raw'], P)
    assert not net.data_dep_init_done
    z, ld = net.forward(g['x_first'])
    assert net.data_dep_init_done
    assert rel(cpu(z), g['z_first']) < 5e-5 and rel(cpu(ld), g['ld_first']) < 5e-5
    used = net.used_mask()
    assert np.max(np.abs(net.store_packed() - g['w_init'])[used]) < 3e-5
    x = g['x']
    for tag in ('init', 'trained'):
        net.load_packed(g['w_' + tag], P)
        net.data_dep_init_done = True
        z, ld = net.forward(x)
        assert rel(cpu(z), g['z_' + tag]) < 2e-5 and rel(cpu(ld), g['ldf_' + tag]) < 2e-5
        xb, ldi = net.inverse(g['z_' + tag])
        assert rel(cpu(xb), g['xb_' + tag]) < 3e-5 and rel(cpu(ldi), g['ldi_' + tag]) < 3e-5
        assert rel(cpu(net.log_probs(x)), g['lp_' + tag]) < 3e-5
        dz = torch.randn_like(z) * 0.01
        dz[:, :S] = 0.0
        x0, _ = net.inverse(z)
        xp, _ = net.inverse(z + dz)
        assert torch.equal(x0[:, :S], xp[:, :S])
    # gradient of the first recorded step and two epochs of Adam
    net.load_packed(g['w_init'], P)
    net.data_dep_init_done = True
    X, jitter = g['X'], float(g['jitter'])
    data = X[g['perms'][0][:100]] + np.float32(jitter) * g['noises'][0][:100]
    loss, grads = net.loss_grad(data)
    assert abs(float(loss) - g['losses'][0]) < 3e-5 * (1 + abs(g['losses'][0]))
    gref, gour = g['grads'][0], net.reference_gradient(grads)
    assert np.max(np.abs(gour - gref)[used]) < 2e-4 * (1e-3 + np.max(np.abs(gref)))
    res = net.train_epochs(X, X[:23], torch.from_numpy(g['perms'].astype(np.int64)), torch.from_numpy(g['noises']), jitter=jitter,
                           batch=100, max_epochs=2, patience=50)
    losses = res['losses'].numpy()[:2, 0] * X.shape[0]
    np.testing.assert_allclose(losses, g['losses'].reshape(2, -1).sum(axis=1), rtol=1e-4)


def test_reference_test_spline_slow():
    """tests/test_flows.py:120-144 restated on this build's Trainer"""
    from nnest_amd.trainer import Trainer
    for num_slow in [2, 5]:
        for num_fast in [2, 3, 4, 5]:
            dims = num_slow + num_fast
            t = Trainer(dims, num_slow=num_slow, flow='spline', log_dir=None, log_level=30)
            assert type(t.netG).__name__ == 'HipFastSlowSpline'
            test_data = torch.from_numpy(np.random.normal(size=(10, dims))).float()
            z, z_log_det = t.forward(test_data)
            x, x_log_det = t.inverse(z)
            assert abs(float(torch.max(x.cpu() - test_data))) <= 1e-5
            assert abs(float(torch.max(x_log_det + z_log_det))) <= 1e-5
            dz = torch.randn_like(z) * 0.01
            dz[:, 0:num_slow] = 0.0
            xp, _ = t.inverse(z + dz)
            assert float(torch.max((x - xp)[:, :num_slow].abs())) == 0
            assert t.get_synthetic_samples(10).shape == torch.Size([10, dims])
            assert t.log_probs(test_data).shape == torch.Size([10])
