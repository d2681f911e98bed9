"""GPU parity for the 'choleksy' flow (reference SingleSpeedCholeksy, networks.py:162-239): passes, every gradient element and
Adam steps against fixtures produced by the reference (tests/golden/cholesky_*.npz), and a nested-sampling run on it."""
import glob
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')
from oracle import oracle as orc  # noqa: E402  (checker only)

G = os.path.join(os.path.dirname(__file__), 'golden')
FILES = sorted(glob.glob(os.path.join(G, 'cholesky_*.npz')))


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / (1.0 + np.abs(b))))


def cpu(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize('path', FILES, ids=[os.path.basename(p)[9:-4] for p in FILES])
def test_cholesky_vs_reference_fixture(path):
    from nnest_amd.cholesky import HipCholesky
    if not torch.cuda.is_available():
        pytest.fail('no GPU visible: the -m gpu tests must run on an MI355X')
    g = np.load(path)
    D = int(g['D'])
    net = HipCholesky(D)
    assert list(net.state_dict().keys()) == [str(k) for k in g['keys']]
    ident, _ = net.forward(g['x'])
    assert rel(cpu(ident), g['x']) < 2e-6                 # identity initialisation (networks.py:183-189)
    net.load_packed(g['w0'])
    z, ld = net.forward(g['x'])
    assert rel(cpu(z), g['z']) < 1e-5 and rel(cpu(ld), g['ldf']) < 1e-5
    xb, ldi = net.inverse(g['z'])
    assert rel(cpu(xb), g['xb']) < 1e-5 and rel(cpu(ldi), g['ldi']) < 1e-5
    assert rel(cpu(net.log_probs(g['x'])), g['lp']) < 2e-5
    o = orc.Cholesky(D, g['w0'])
    assert rel(cpu(net.inverse(g['z'])[0]), o.inverse(g['z'], f64=True)[0]) < 1e-5
    X, jitter = g['X'], float(g['jitter'])
    data = X[g['perms'][0][:100]] + np.float32(jitter) * g['noises'][0][:100]
    loss, grad = net.loss_grad(data)
    assert abs(float(loss) - g['losses'][0]) < 2e-5 * (1 + abs(g['losses'][0]))
    assert np.max(np.abs(cpu(grad) - g['grads'][0])) < 5e-5 * (1e-3 + np.max(np.abs(g['grads'][0])))
    res = net.train_epochs(X, X[:23], torch.from_numpy(g['perms'].astype(np.int64)), torch.from_numpy(g['noises']), jitter=jitter,
                           batch=100, max_epochs=2, patience=50)
    np.testing.assert_allclose(res['losses'].numpy()[:2, 0] * X.shape[0], g['losses'].reshape(2, -1).sum(axis=1), rtol=3e-5)
    if res['best_epoch'] == 2:
        dref, dour = g['ws'][-1] - g['w0'], net.store_packed() - g['w0']
        assert np.sqrt(np.mean((dour - dref) ** 2)) < 0.05 * np.sqrt(np.mean(dref ** 2))


def test_nested_run_on_the_cholesky_flow(tmp_path):
    from nnest_amd.likelihoods import Gaussian
    from nnest_amd.nested import NestedSampler
    np.random.seed(3)
    torch.manual_seed(3)
    like = Gaussian(2, 0.9, lim=3)
    s = NestedSampler(2, like, transform=lambda x: 3 * x, log_dir=str(tmp_path), num_live_points=300, log_level=30, flow='choleksy')
    assert type(s.trainer.netG).__name__ == 'HipCholesky' and s._fused_like_id is None
    s.run(mcmc_num_chains=20, train_iters=100, mcmc_steps=10)
    assert abs(s.logz - math.log(1 / 36.0)) <= 0.5, s.logz   # unit-mass Gaussian inside [-3, 3]^2 (up to ~1 % truncation)
