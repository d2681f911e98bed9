"""GPU parity of the masked autoregressive flow (nnest_maf_create; nnest_amd/csrc/maf_tile.h, maf_kernels.h, maf_train.h;
SURVEY.md 8 row a22; BASELINE config 5 names it).

[UNPINNED: absent from the reference -- nnest/trainer.py:83-100 dispatches 'choleksy' / 'nvp' / 'spline' only.]  Every test
compares the HIP path with the oracle's restatement of the same build-defined flow (oracle/maf_oracle_impl.h, itself
checked for self-consistency on the CPU: tests/test_oracle_maf.py) and with the reference's own criteria for a flow
(tests/test_flows.py:8, :27-30: round trip <= 1e-5, log-det antisymmetry)."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc  # checker only
from tests.mh_checks import assert_borderline, first_divergence

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from nnest_amd import maf
    if not torch.cuda.is_available():
        pytest.fail('no GPU visible: the -m gpu tests must run on an MI355X')
    return maf


def cpu(t):
    return t.detach().cpu().numpy()


def rel(a, b):
    return float(np.max(np.abs(a - b) / (1.0 + np.abs(b))))


def pair(hip, D, B=3, L=1, seed=0, scale=1.0):
    m = hip.HipMAF(D, 16, B, L, seed=seed)
    if scale != 1.0:
        m.load_packed(m.store_packed() * np.float32(scale))
    return m, orc.NVP(D, 16, B, L, m.store_packed(), kind='maf')


@pytest.mark.parametrize('D,B,L', [(2, 3, 1), (5, 3, 1), (50, 3, 1), (100, 3, 1), (33, 2, 0), (64, 4, 2), (128, 1, 1)])
@pytest.mark.parametrize('N', [1, 37, 1000])
def test_maf_passes_vs_oracle(hip, D, B, L, N):
    """[UNPINNED]  forward (one pass), inverse (group by group), log_probs against the oracle; the reference's flow criteria
    (tests/test_flows.py:27-30): round trip <= 1e-5, log-det antisymmetry"""
    m, o = pair(hip, D, B, L, seed=D)
    assert m.num_groups == orc.lib().orc_maf_num_groups(D, 16)
    x = np.random.RandomState(N + D).uniform(-1, 1, size=(N, D)).astype(np.float32)
    z, ld = m.forward(x)
    zo, ldo = o.forward(x)
    assert rel(cpu(z), zo) < 2e-5 and rel(cpu(ld), ldo) < 2e-5
    xb, ldi = m.inverse(z)
    assert float(torch.max(torch.abs(xb - torch.from_numpy(x).cuda()))) <= 1e-5
    assert float(torch.max(torch.abs(ld + ldi))) <= 1e-4 * max(1.0, float(ld.abs().max()))
    xo, ldio = o.inverse(cpu(z))
    assert rel(cpu(xb), xo) < 2e-5 and rel(cpu(ldi), ldio) < 3e-5
    assert rel(cpu(m.log_probs(x)), o.log_probs(x)) < 3e-5
    zn = np.random.RandomState(1).standard_normal((N, D)).astype(np.float32)    # sampling: inverse of base draws
    xs, _ = m.inverse(zn)
    xso, _ = o.inverse(zn)
    assert rel(cpu(xs), xso) < 5e-5


def test_maf_empty_and_ragged(hip):
    """[UNPINNED]  N = 0 and sizes around the 16-row tiles"""
    m, o = pair(hip, 20, seed=1)
    z, ld = m.forward(np.zeros((0, 20), np.float32))
    assert z.shape == (0, 20) and ld.shape == (0,)
    for N in (15, 16, 17, 4097):
        x = np.random.RandomState(N).uniform(-1, 1, size=(N, 20)).astype(np.float32)
        z, ld = m.forward(x)
        zo, ldo = o.forward(x)
        assert rel(cpu(z), zo) < 2e-5 and rel(cpu(ld), ldo) < 2e-5


@pytest.mark.parametrize('D,like,scale', [(2, 'rosenbrock', 5.0), (20, 'gaussmix', 10.0), (50, 'rosenbrock', 5.0), (100, 'rosenbrock', 5.0)])
def test_maf_fused_eval_vs_oracle(hip, D, like, scale):
    """[UNPINNED]  K3 with the MAF: x = f^-1(z), box prior, likelihood in one launch"""
    m, o = pair(hip, D, seed=3)
    z = (0.4 * np.random.RandomState(D).standard_normal((300, D))).astype(np.float32)
    x, ld, logl, inbox = m.inverse_loglike(hip._lib.LIKE_IDS[like], scale, z)
    xo, ldo = o.inverse(z)
    assert rel(cpu(x), xo) < 3e-5 and rel(cpu(ld), ldo) < 3e-5
    assert np.array_equal(cpu(inbox) == 1, orc.prior_inbox(cpu(x)) == 0)
    np.testing.assert_allclose(cpu(logl), orc.loglike(like, cpu(x), scale), rtol=3e-6, atol=1e-5)


@pytest.mark.parametrize('D,B,L,M', [(2, 3, 1, 100), (5, 3, 1, 37), (50, 3, 1, 100), (100, 3, 1, 100), (33, 2, 0, 128), (20, 2, 2, 16)])
def test_maf_loss_and_every_gradient_element_vs_oracle(hip, D, B, L, M):
    """[UNPINNED]  loss = -mean(log_probs) (trainer.py:394) and dloss/dw of one minibatch: every element against the oracle's
    reverse mode (itself checked against finite differences, tests/test_oracle_maf.py); masked parameters exactly zero"""
    m, o = pair(hip, D, B, L, seed=5)
    X = np.random.RandomState(M).uniform(-1, 1, size=(M, D)).astype(np.float32)
    loss, grad = m.loss_grad(X)
    lo, go = o.loss_grad(X)
    g = cpu(grad)
    assert abs(float(loss[0]) - lo) < 2e-5 * max(1.0, abs(lo))
    assert np.max(np.abs(g - go)) < 2e-4 * np.max(np.abs(go)), float(np.max(np.abs(g - go)) / np.max(np.abs(go)))
    ns = m.num_params // (2 * B)
    live = np.array([orc.lib().orc_maf_param_live(D, 16, L, b, i) for b in range(B) for _ in range(2) for i in range(ns)], dtype=bool)
    assert np.all(g[~live] == 0) and np.count_nonzero(g[live]) > 0.5 * live.sum()
    _, g2 = m.loss_grad(X)
    assert torch.equal(grad, g2)    # one producer per element, fixed summation order: bitwise reproducible


def test_maf_adam_steps_follow_the_oracle(hip):
    """[UNPINNED]  five minibatch steps of Trainer._train (trainer.py:384-403) -- gradient, torch.optim.Adam with coupled weight
    decay (trainer.py:121-122), refreshed fragment images -- against the oracle's train_step on the same rows"""
    D = 20
    m, o = pair(hip, D, seed=9)
    rng = np.random.RandomState(0)
    X = rng.uniform(-1, 1, size=(500, D)).astype(np.float32)
    for k in range(5):
        idx = rng.permutation(500)[:100].astype(np.int32)
        loss, grad = m.loss_grad(X[idx])
        m.adam_step(grad, 1e-3, 1e-6)
        lo, _ = o.train_step(X, idx, None, 0.0, lr=1e-3, wd=1e-6)
        assert abs(float(loss[0]) - lo) < 3e-5 * max(1.0, abs(lo))
    assert np.max(np.abs(m.store_packed() - o.w)) < 2e-6
    mm, vv = m.adam_moments()
    assert np.max(np.abs(mm - o.m)) < 1e-6 * max(1.0, np.max(np.abs(o.m))) + 1e-8 and m.adam_step_count() == 5
    z, _ = m.forward(X[:64])           # the images follow the stepped weights
    zo, _ = o.forward(X[:64])
    assert rel(cpu(z), zo) < 2e-5


@pytest.mark.parametrize('D,L', [(7, 1), (50, 1), (100, 2)])
def test_maf_epoch_call_equals_the_stepwise_loop(hip, D, L):
    """[UNPINNED]  nnest_maf_train_epoch (gradient kernel + ONE kernel that reduces, steps Adam and rewrites both fragment images
    through position maps) against the same minibatches driven step by step (nnest_nvp_loss_grad + nnest_nvp_adam_step, whose
    images are rebuilt by the gather kernel): the same weights, moments and step count bit for bit, the same running loss, and
    passes that follow the new weights."""
    import ctypes
    from nnest_amd import _lib
    a, _ = pair(hip, D, 3, L, seed=4, scale=0.5)
    b = hip.HipMAF(D, 16, 3, L, seed=4)
    b.load_packed(a.store_packed())
    rng = np.random.RandomState(D)
    X = torch.from_numpy((rng.standard_normal((230, D)) * 0.3).astype(np.float32)).cuda()
    tot = torch.zeros((), dtype=torch.float32, device='cuda')
    _lib.check(a._lib.nnest_maf_train_epoch(a._h, _lib.ptr(X), 230, 100, ctypes.c_float(1e-3), ctypes.c_float(1e-6), _lib.ptr(tot),
                                            _lib.current_stream(a.device)))
    ref = 0.0
    for b0 in range(0, 230, 100):
        loss, grad = b.loss_grad(X[b0:b0 + 100])
        b.adam_step(grad, 1e-3, 1e-6)
        ref += float(loss[0])
    assert np.array_equal(a.store_packed(), b.store_packed())
    ma, va = a.adam_moments()
    mb, vb = b.adam_moments()
    assert np.array_equal(ma, mb) and np.array_equal(va, vb) and a.adam_step_count() == b.adam_step_count() == 3
    assert abs(float(tot) - ref) <= 1e-6 * max(1.0, abs(ref))
    za, lda = a.forward(X[:64])      # the images written by the update kernel == the images rebuilt from the weights
    zb, ldb = b.forward(X[:64])
    assert torch.equal(za, zb) and torch.equal(lda, ldb)
    xa, _ = a.inverse(za)
    xb, _ = b.inverse(zb)
    assert torch.equal(xa, xb)
    la, ga = a.loss_grad(X[:100])    # ... the transposed image too (the backward pass reads it)
    lb, gb = b.loss_grad(X[:100])
    assert torch.equal(ga, gb) and torch.equal(la, lb)


@pytest.mark.parametrize('D,C,S,dyn', [(50, 200, 12, False), (100, 64, 6, False), (5, 37, 30, True), (20, 1000, 8, True)])
def test_maf_metropolis_kernel_vs_oracle(hip, D, C, S, dyn):
    """[UNPINNED]  K4 (Sampler._mcmc_sample, sampler.py:229-463) with the MAF's grouped inverse inside the persistent kernel:
    the kernel's own noise replayed through the oracle, per 16-walker adaptation group"""
    m, o = pair(hip, D, seed=2)
    rng = np.random.RandomState(C)
    init = rng.uniform(-0.4, 0.4, size=(C, D))
    init_logl = orc.loglike('rosenbrock', init, 5.0)
    loglstar = float(np.min(init_logl)) - 1e3
    dz, u = m.fill_noise(S, C, seed=17, walker_offset=5)
    z, _ = m.forward(init)
    logl = torch.from_numpy(init_logl).cuda()
    res = m.mh_steps(0, 5.0, z, logl, loglstar, 0.05, S, dynamic='group' if dyn else False, seed=17, walker_offset=5, history=True)
    hx, hl = cpu(res['hist_x']), cpu(res['hist_logl'])
    n_bad = 0
    for g0 in range(0, C, 16):
        sl = slice(g0, min(g0 + 16, C))
        margins = np.empty((S, min(g0 + 16, C) - g0))
        so, _, lo, sc, ncall, (acc, rej) = orc.mcmc_sample(o, 'rosenbrock', 5.0, init[sl], init_logl[sl], loglstar, 0.05, dyn,
                                                           cpu(dz)[:, sl], cpu(u)[:, sl], margins=margins)
        if int(res['n_call'][sl].sum()) == ncall and int(res['n_accept'][sl].sum()) == acc:
            assert rel(hx[sl], so) < 1e-4 and rel(hl[sl], lo) < 1e-4
            assert abs(float(res['scale'][g0 // 16]) - sc) < 1e-5 * max(1.0, sc)
        else:
            n_bad += 1
            first = first_divergence(hx[sl], so)
            w = min((int(s_), int(k)) for k, s_ in enumerate(first) if s_ >= 1)[1]
            assert_borderline(hx[sl], so, margins, [w], tol=3e-4)
    assert n_bad <= max(1, (C // 16) // 10) and int(res['n_accept'].sum()) > 0
    xo, _ = o.inverse(cpu(z))
    assert rel(cpu(res['x']), xo) < 1e-4           # the final x is f^-1 of the final latent
    assert m.mh_form_for(C) == 'image' and m.mh_form_for(C, form='solo') is None


def test_maf_trainer_and_nested_run(hip, tmp_path):
    """[UNPINNED]  Trainer(flow='maf') behind the reference's Trainer protocol (trainer.py:134-301): training lowers the loss;
    NestedSampler(flow='maf') reproduces the reference's integration criterion (tests/test_nested.py:10-19: Rosenbrock 2-D,
    |logZ + 5.80| within the statistical error)"""
    from nnest_amd.trainer import Trainer
    from nnest_amd.nested import NestedSampler
    from nnest_amd.likelihoods import Rosenbrock
    np.random.seed(0)
    torch.manual_seed(0)
    rng = np.random.RandomState(0)
    A = rng.standard_normal((6, 6)) * 0.2 + 0.4 * np.eye(6)
    X = np.clip(rng.standard_normal((800, 6)) @ A.T * 0.3, -1, 1)
    tr = Trainer(6, flow='maf', log_dir=None, learning_rate=2e-3, seed=1)
    before = float(-tr.netG.log_probs(X).mean())
    tr.train(X, max_iters=30, jitter=0.01)
    after = float(-tr.netG.log_probs(X).mean())
    assert after < before - 0.5 and tr.best_validation_epoch >= 1 and tr.losses.shape[1] == 2
    assert set(tr.netG.state_dict()) == set(k for k, _, _ in tr.netG.layer_shapes())
    s = NestedSampler(2, Rosenbrock(2), transform=lambda x: 5 * x, log_dir=str(tmp_path), num_live_points=400, log_level=30,
                      flow='maf')
    assert s._fused_like_id is not None
    s.run(train_iters=50, mcmc_num_chains=100)
    assert abs(s.logz + 5.804) <= 0.35, s.logz


def test_config5_slice_with_the_maf(hip, tmp_path):
    """[UNPINNED]  BASELINE config 5 as named -- Rosenbrock x_dim 100, 8000 live points, MAF flow -- as a bounded slice beside
    the RealNVP one (tests/test_gpu_nested.py): one retrain, fused proposal batches of 8000 walkers x 100 steps"""
    from nnest_amd.nested import NestedSampler
    from nnest_amd.likelihoods import Rosenbrock
    np.random.seed(0)
    torch.manual_seed(0)
    s = NestedSampler(100, Rosenbrock(100), transform=lambda x: 5 * x, log_dir=str(tmp_path), num_live_points=8000, log_level=30,
                      flow='maf')
    assert s._fused_like_id is not None and s.trainer.netG.num_groups == 17
    s.run(strategy=['mcmc'], mcmc_num_chains=8000, mcmc_steps=100, max_iters=3000, train_iters=2)
    assert s.niter >= 3000 and s.num_retrains == 1 and s.num_batches >= 1
    assert np.isfinite(s.logz) and np.all(np.diff(s.loglikes[:3000]) >= 0)
