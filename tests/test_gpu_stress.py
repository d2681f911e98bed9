"""A cut of the stress tools inside `pytest -m gpu` (round-4 verdict: tools/stress_k4_sync.py and tools/stress_k5_shapes.py found a
latent inline-asm register-liveness hazard -- one launch in 25 lost a window word and a bounded wait gave up -- and lived outside the
driver's GPU test run, so a regression of that class would have passed it).  Repeated launches of the persistent kernels with
their in-kernel cross-CU waits: no wait may run out, and the same launch must give the same bits every time."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from nnest_amd import flow
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return flow


@pytest.mark.parametrize('which,kw', [('product rule: 16 exact steps, then 8 behind', dict(form='solo', dynamic='batch', lag=8, warm=16)),
                                      ('4 exact steps, then 8 behind', dict(form='solo', dynamic='batch', lag=8, warm=4)),
                                      ('fixed step', dict(form='solo'))])
def test_k4_repeated_launches_never_run_out_of_a_wait(hip, which, kw):
    """nnest/sampler.py:229-463 as one launch, 70 launches per step rule at config 2's population (210 in all): the sync buffer's
    error word stays 0 (it says which wait, step and workgroup otherwise), no launch takes 10x the usual time, and a repeated seed
    reproduces its chain bit for bit."""
    D, C, S = 50, 1000, 250
    nvp = hip.HipNVP(D, 16, 3, 1, seed=0)
    u0 = np.random.RandomState(0).uniform(-1, 1, size=(C, D))
    z0, _ = nvp.forward(u0)
    l0 = hip.loglike(0, u0, 5.0)
    star, step = float(l0.min()), 1 / np.sqrt(D)
    first = None
    for k in range(70):
        z, l = z0.clone(), l0.clone()
        res = nvp.mh_steps(0, 5.0, z, l, star, step, S, seed=k % 35, **kw)
        v = int(res['sync'][-1].item()) & 0xffffffff if res.get('sync') is not None else 0
        assert v == 0, '%s, launch %d: wait code %d at step %d of workgroup %d' % (which, k, v & 255, (v >> 8) & 4095, v >> 20)
        if k == 0:
            first = (res['x'].clone(), l.clone(), res['n_accept'].clone())
        if k == 35:   # seed 0 again
            assert torch.equal(res['x'], first[0]) and torch.equal(l, first[1]) and torch.equal(res['n_accept'], first[2])


@pytest.mark.parametrize('D,N,E', [(50, 1000, 12), (20, 2000, 6), (100, 1000, 6), (5, 333, 12), (50, 777, 10)])
def test_k5_repeated_launches_never_run_out_of_a_wait_and_repeat_their_bits(hip, D, N, E):
    """nnest/trainer.py:198-241 as one launch (train_kernel_rows: grid barriers, tagged weight publish), 8 launches per shape from the
    same state: none reports a wait that ran out (train_epochs raises then), all leave the same weights bit for bit -- the property
    replicated training across ranks relies on."""
    rng = np.random.RandomState(N)
    live = rng.uniform(-1, 1, size=(N, D))
    nv = max(N // 10, 1)
    perms = torch.stack([torch.randperm(N - nv, generator=torch.Generator().manual_seed(e)) for e in range(E)]).int()
    ref = None
    for k in range(8):
        nvp = hip.HipNVP(D, 16, 3, 1, seed=7)
        nvp.train_epochs(live[nv:], live[:nv], perms, None, max_epochs=E, seed=3, jitter=0.01, batch=100, patience=1000)
        w = np.asarray(nvp.store_packed())
        if ref is None:
            ref = w
        assert np.array_equal(ref, w), 'launch %d: weights differ from the first launch (max %.3g)' % (k, np.abs(ref - w).max())


def test_a_refused_launch_does_not_hand_the_next_one_dirty_sync_words(hip):
    """the double-buffered sync words (flow.mh_steps) changed halves BEFORE the launch was known to have run; a launch the
    library refuses (here: exact warm-up steps asked of the quad form, NNEST_E_UNSUPPORTED -- what sampler.py's fallback path does)
    zeroed nothing, and the next launch of the same step count got the half that was dirty from two launches ago.  Successful
    launches, a refused one, successful ones again: no wait runs out, and the chains are those of a flow that never saw the refusal."""
    D, C, S = 50, 1000, 60
    u0 = np.random.RandomState(0).uniform(-1, 1, size=(C, D))
    l0 = hip.loglike(0, u0, 5.0)
    star, step = float(l0.min()), 1 / np.sqrt(D)
    kw = dict(form='solo', dynamic='batch', lag=8, warm=16)

    def launch(nvp, seed, **over):
        z, _ = nvp.forward(u0)
        l = l0.clone()
        res = nvp.mh_steps(0, 5.0, z, l, star, step, S, seed=seed, **dict(kw, **over))
        hip.HipNVP.check_sync(res)
        return res['x'].clone(), l, res['n_accept'].clone(), float(res['scale'][0].item())

    clean = hip.HipNVP(D, 16, 3, 1, seed=0)
    want = [launch(clean, k) for k in range(6)]
    nvp = hip.HipNVP(D, 16, 3, 1, seed=0)
    got = [launch(nvp, k) for k in range(3)]
    for _ in range(2):   # (an odd and an even number of refusals in a row)
        with pytest.raises(hip._lib.NnestHipError):
            launch(nvp, 99, form='quad')
        got.append(launch(nvp, len(got)))
    with pytest.raises(hip._lib.NnestHipError):
        launch(nvp, 99, form='quad')
    got.append(launch(nvp, len(got)))
    assert len(got) == 6
    for a, b in zip(got, want):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and a[3] == b[3]
