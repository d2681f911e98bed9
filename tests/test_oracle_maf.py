"""CPU checks of the oracle's masked autoregressive flow (oracle/maf_oracle_impl.h).

[UNPINNED: absent from the reference -- nnest/trainer.py:83-100 knows only 'choleksy' / 'nvp' / 'spline'.]  The flow is
build-defined, so what is tested is self-consistency, the way the reference checks its own flows (tests/test_flows.py:27-30:
round trip <= 1e-5 and log-det antisymmetry; trainer.py:373-382: log-det against the Jacobian): round trip, antisymmetry,
log|det J| by brute force in float64, the autoregressive structure of J, the group-by-group inverse against the textbook
one-dimension-at-a-time inverse, the analytic gradient against finite differences, masked parameters' zero gradient."""
import ctypes

import numpy as np
import pytest

from oracle import oracle as orc

SHAPES = [(2, 16, 3, 1), (5, 16, 3, 1), (8, 16, 2, 0), (7, 32, 3, 2), (50, 16, 3, 1), (100, 16, 3, 1), (20, 64, 2, 1)]


def make(D, H, B, L, seed=0, scale=0.35):
    rng = np.random.RandomState(seed)
    m = orc.NVP(D, H, B, L, kind='maf')
    m.w[:] = (scale * rng.standard_normal(m.n) / np.sqrt(H)).astype(np.float32)
    return m, rng


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


@pytest.mark.parametrize('D,H,B,L', SHAPES)
def test_maf_round_trip_and_logdet_antisymmetry(D, H, B, L):
    """[UNPINNED]  the reference's own criterion for a flow (tests/test_flows.py:8, :27-30)"""
    m, rng = make(D, H, B, L)
    x = rng.uniform(-1, 1, size=(40, D))
    z, ld = m.forward(x)
    xb, ldi = m.inverse(z)
    assert np.max(np.abs(xb - x)) <= 1e-5
    assert np.max(np.abs(ld + ldi)) <= 1e-4 * max(1.0, np.max(np.abs(ld)))
    z64, ld64 = m.forward(x, f64=True)
    xb64, ldi64 = m.inverse(z64, f64=True)
    assert np.max(np.abs(xb64 - x)) <= 1e-11 and np.max(np.abs(ld64 + ldi64)) <= 1e-10
    assert np.max(np.abs(z - z64)) < 1e-4 and np.max(np.abs(ld - ld64)) < 1e-3
    lp = m.log_probs(x, f64=True)
    assert np.allclose(lp, -0.5 * np.sum(z64 ** 2, axis=1) - 0.5 * D * np.log(2 * np.pi) + ld64, rtol=0, atol=1e-9)


@pytest.mark.parametrize('D,H,B,L', [(2, 16, 3, 1), (5, 16, 3, 1), (8, 16, 2, 0), (7, 32, 3, 2), (12, 16, 1, 1)])
def test_maf_logdet_is_the_jacobian_and_the_jacobian_is_autoregressive(D, H, B, L):
    """[UNPINNED]  log|det dz/dx| by central differences in float64 (trainer.py:373-382 checks its flows this way); and
    with ONE block the Jacobian is lower triangular in the block's degree order (that is what "autoregressive" means)"""
    m, rng = make(D, H, B, L, seed=3)
    x0 = rng.uniform(-0.8, 0.8, size=D)
    h = 1e-6
    J = np.empty((D, D))
    for j in range(D):
        e = np.zeros(D); e[j] = h
        zp, _ = m.forward((x0 + e)[None], f64=True)
        zm, _ = m.forward((x0 - e)[None], f64=True)
        J[:, j] = (zp[0] - zm[0]) / (2 * h)
    _, ld = m.forward(x0[None], f64=True)
    sign, logabs = np.linalg.slogdet(J)
    assert abs(logabs - ld[0]) < 1e-6 and sign > 0
    if B == 1:
        assert np.max(np.abs(np.triu(J, 1))) < 1e-9          # even block: natural order -> lower triangular
        assert np.min(np.abs(np.diag(J))) > 0


@pytest.mark.parametrize('D,H,B,L', SHAPES)
def test_maf_grouped_inverse_equals_one_dimension_at_a_time(D, H, B, L):
    """[UNPINNED]  G <= H + 1 passes instead of D: the same numbers, bit for bit (the masked weights multiply the
    not-yet-final coordinates by exact zeros)"""
    m, rng = make(D, H, B, L, seed=5)
    z = rng.standard_normal((16, D))
    G = orc.lib().orc_maf_num_groups(D, H)
    assert 2 <= G <= min(H, D - 1) + 1
    groups = [orc.lib().orc_maf_group(D, H, 0, d) for d in range(D)]
    assert groups == sorted(groups) and groups[0] == 0 and max(groups) == G - 1 and len(set(groups)) == G
    assert [orc.lib().orc_maf_group(D, H, 1, d) for d in range(D)] == groups[::-1]    # odd blocks: the order reversed
    x, ld = m.inverse(z, f64=True)
    xs, lds = np.empty_like(x), np.empty_like(ld)
    z64 = np.ascontiguousarray(z, dtype=np.float64)
    orc.lib().orc64_maf_inverse_seq(*m._cfg(), _dp(z64), z.shape[0], _dp(xs), _dp(lds))
    assert np.array_equal(x, xs) and np.allclose(ld, lds, rtol=0, atol=1e-12)   # (the log-det terms are added in another order)


@pytest.mark.parametrize('D,H,B,L', [(2, 16, 3, 1), (5, 16, 2, 1), (7, 32, 2, 2), (9, 16, 3, 0)])
def test_maf_gradient_against_finite_differences(D, H, B, L):
    """[UNPINNED]  loss = -mean(log_probs): analytic reverse mode against central differences (float64) on the parameters
    the masks leave alive; the masked ones have gradient exactly zero and do not influence the loss"""
    m, rng = make(D, H, B, L, seed=7)
    X = rng.uniform(-1, 1, size=(9, D)).astype(np.float32)
    loss, g = m.loss_grad(X, f64=True)
    ns = m.n // (2 * B)
    live = np.array([orc.lib().orc_maf_param_live(D, H, L, b, i) for b in range(B) for _ in range(2) for i in range(ns)], dtype=bool)
    assert np.all(g[~live] == 0) and 0.3 < live.mean() < 0.95
    idx = np.concatenate([rng.choice(np.flatnonzero(live), 60), rng.choice(np.flatnonzero(~live), 10)])
    kinks = 0
    for i in idx:
        w0 = m.w[i]
        h = 2.5e-4
        m.w[i] = w0 + h
        lp = -float(np.mean(m.log_probs(X, f64=True)))
        m.w[i] = w0 - h
        lm = -float(np.mean(m.log_probs(X, f64=True)))
        m.w[i] = w0
        fd = (lp - lm) / (float(np.float32(w0 + h)) - float(np.float32(w0 - h)))   # the weights are float32: the step actually taken
        ok = abs(fd - g[i]) < 5e-5 + 1e-3 * abs(g[i])
        if not ok and (i // ns) % 2 == 1:   # a ReLU of the translate net switching inside +-h: the difference quotient is not the derivative
            kinks += 1
            continue
        assert ok, (i, fd, g[i], live[i])
    assert kinks <= 3
    _, g32 = m.loss_grad(X)
    assert np.max(np.abs(g32 - g)) < 2e-4 * max(1.0, np.max(np.abs(g)))


def test_maf_trains_and_proposes():
    """[UNPINNED]  Adam steps lower the loss on a correlated Gaussian, and the Metropolis loop of the oracle
    (sampler.py:229-463 restated) runs on the MAF through the same entry points"""
    D = 6
    m, rng = make(D, 16, 3, 1, seed=11, scale=0.1)
    A = rng.standard_normal((D, D)) * 0.3 + np.eye(D) * 0.5
    X = (rng.standard_normal((600, D)) @ A.T).astype(np.float32)
    first = m.valid_loss(X)
    for ep in range(12):
        perm = rng.permutation(500).astype(np.int32)
        for k in range(5):
            m.train_step(X[:500], perm[100 * k:100 * (k + 1)], None, 0.0, lr=3e-3)
    assert m.valid_loss(X) < first - 0.3
    init = rng.uniform(-0.3, 0.3, size=(8, D))
    il = orc.loglike('rosenbrock', init, 5.0)
    dz = rng.standard_normal((6, 8, D)).astype(np.float32)
    u = rng.uniform(size=(6, 8)).astype(np.float32)
    s, lat, ll, sc, ncall, (acc, rej) = orc.mcmc_sample(m, 'rosenbrock', 5.0, init, il, -1e300, 0.05, False, dz, u)
    assert acc + rej == 48 and np.all(np.isfinite(s))
    xb, _ = m.inverse(lat[:, -1])
    assert np.max(np.abs(xb - s[:, -1])) < 1e-5
