"""[UNPINNED] CPU checks of the checker of the build-defined slice proposal (oracle/oracle.py::slice_sample; the reference has no
slice proposal, nnest/sampler.py:310-316): the restatement keeps the invariants of a slice-sampling update under a hard constraint,
its uniforms are the Philox words the kernel draws (Random123's known-answer vector pins the generator), and it is a function of
its arguments."""
import numpy as np

from oracle import oracle as orc


def test_philox_known_answer_and_the_uniform_built_on_it():
    # Random123 kat_vectors: philox4x32-10, counter = key = 0 / all ones
    assert orc.philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert orc.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    u = [float(orc.slice_uniform(7, w, k)) for w in range(3) for k in range(64, 70)]
    assert all(0.0 <= v < 1.0 for v in u) and len(set(u)) == len(u)
    assert orc.slice_uniform(7, 1, 64) == orc.slice_uniform(7, 1, 64) and orc.slice_uniform(7, 1, 64) != orc.slice_uniform(8, 1, 64)


def test_slice_sample_keeps_the_constraint_and_moves():
    D, C, S = 4, 12, 6
    rng = np.random.RandomState(0)
    nvp = orc.NVP(D, 16, 3, 1, (rng.uniform(-1, 1, size=orc.NVP(D, 16, 3, 1).n) * 0.3).astype(np.float32))
    u0 = rng.uniform(-0.5, 0.5, size=(C, D))
    l0 = orc.loglike('rosenbrock', u0, 5.0)
    star = float(np.sort(l0)[2])
    keep = l0 > star
    u0, l0 = u0[keep], l0[keep]
    z0, _ = nvp.forward(u0.astype(np.float32))
    dz = rng.standard_normal((S, u0.shape[0], D)).astype(np.float32)
    margins = np.empty((S, u0.shape[0]))
    a = orc.slice_sample(nvp, 'rosenbrock', 5.0, z0, l0, star, 0.7, dz, seed=3, walker_offset=10, margins=margins)
    assert np.all(np.abs(a['x'][:, -1]) <= 1.0) and np.all(a['logl'] > star)
    assert np.all(a['n_eval'] >= a['n_call']) and np.all(a['n_call'] >= a['n_move']) and a['n_move'].mean() > 0.9 * S
    np.testing.assert_allclose(a['logl'], orc.loglike('rosenbrock', a['x'][:, -1], 5.0), rtol=1e-6, atol=1e-6)
    xb, _ = nvp.inverse(a['z'])
    assert np.max(np.abs(xb - a['x'][:, -1])) < 1e-5
    b = orc.slice_sample(nvp, 'rosenbrock', 5.0, z0, l0, star, 0.7, dz, seed=3, walker_offset=10)
    assert np.array_equal(a['x'], b['x']) and np.array_equal(a['n_eval'], b['n_eval'])
    c = orc.slice_sample(nvp, 'rosenbrock', 5.0, z0, l0, star, 0.7, dz, seed=4, walker_offset=10)
    assert not np.array_equal(a['x'], c['x'])
    assert np.all(np.isfinite(margins)) and np.all(margins >= 0)


def test_exact_rosenbrock_evidence_by_transfer_quadrature():
    """oracle/rosenbrock_exact.py (the known answer the slice proposal's convergence is judged by): equals the 2-D closed
    form the nested tests use (-5.804), a brute-force 3-D grid sum of the oracle's own log-likelihood, and is converged in the grid step."""
    from oracle.rosenbrock_exact import log_evidence
    assert abs(log_evidence(2) + 5.804132) < 1e-5
    h = 0.04
    g = np.arange(-5 + h / 2, 5, h)                       # midpoint rule, 250^3 cells
    tot = 0.0
    for x1 in g:
        pts = np.stack(np.meshgrid([x1], g, g, indexing='ij'), axis=-1).reshape(-1, 3) / 5.0
        tot += np.exp(orc.loglike('rosenbrock', pts, 5.0)).sum()
    brute = np.log(tot * h ** 3) - 3 * np.log(10.0)
    assert abs(log_evidence(3) - brute) < 2e-3, (log_evidence(3), brute)
    assert abs(log_evidence(50, h=0.01) - log_evidence(50, h=0.005)) < 1e-6
    assert abs(log_evidence(50) + 231.9384) < 1e-3
