#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING the upstream reference.

Test infrastructure, run in the build container only:  python oracle/gen_golden.py
The reference (/root/reference, adammoss/nnest v0.4.2) is imported unmodified through
oracle/_refimport.py.  Only plain arrays (inputs + the reference's outputs) are written;
no reference source, bytecode or pickled module goes into the fixtures.

Fixture families:
  G1 flow_*.npz       packed NVP weights, x -> (z, logdet) -> (x', logdet'), log_probs
                      (nnest/networks.py:24-42, :71-76, :289-309; nnest/trainer.py:247-301)
  G2 like_*.npz       Rosenbrock / GaussianMix / Himmelblau through safe_loglike
                      (nnest/likelihoods.py:51, :70, :182-189; nnest/sampler.py:110-133)
  G2b like2.npz      Gaussian / Eggbox / GaussianShell / DoubleGaussianShell (nnest/likelihoods.py:77-150)
  G3 prior.npz        UniformPrior box flags through safe_prior (nnest/priors.py:39-43)
  G4 train_*.npz      Trainer._train minibatch steps with recorded shuffle + jitter noise,
                      every gradient, post-Adam weights and moments (nnest/trainer.py:384-418)
  G5 mcmc_*.npz       Sampler._mcmc_sample traces with recorded proposal noise
                      (nnest/sampler.py:229-463)
  G5b mcmc_spline_*.npz  the same on the reference's default flow (flow='spline', nnest/networks.py:458-556, :708-715)
  G6 nested_cfg1.json seeded end-to-end NestedSampler.run on config 1 (nnest/nested.py:97-510)
  G7 trainrun_*.npz   Trainer.train() for a few epochs: split, per-epoch perms, noise, losses
                      (nnest/trainer.py:134-245)
  G9 spline_*.npz     SingleSpeedSpline (nnest/networks.py:393-715): passes before / after ActNorm's data-dependent
                      initialisation, gradients and Adam steps, a trained state
  G10 base_gennormal_*.npz  generalised-normal base distribution (nnest/distributions/generalised_normal.py): log_probs,
                      gradients, Adam steps for both flows
  G11 fastslow_*.npz  FastSlowNVP (nnest/networks.py:86-150, :350-380): passes, gradients, Adam steps
  G8 scale_*.npz      SingleSpeedNVP with scale='translate' / 'constant' (nnest/networks.py:289-347): passes,
                      gradients and Adam steps
"""
import os
import sys
import json
import copy
import shutil
import logging
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _refimport import import_reference  # noqa: E402

import_reference()

import numpy as np  # noqa: E402
import torch  # noqa: E402
from nnest.trainer import Trainer  # noqa: E402
from nnest.nested import NestedSampler  # noqa: E402
from nnest.likelihoods import Rosenbrock, GaussianMix, Himmelblau, Gaussian, Eggbox, GaussianShell, DoubleGaussianShell  # noqa: E402
from nnest.priors import UniformPrior  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
# GOLDEN_ONLY=name1,name2: within the families asked for, write only these cases (new cases are added without re-writing the
# committed fixtures of the old ones)
ONLY = set(filter(None, os.environ.get('GOLDEN_ONLY', '').split(',')))


def wanted(name):
    return not ONLY or name in ONLY

torch.set_num_threads(1)  # single-thread torch so the fixtures do not depend on thread count


def pack_state_dict(netG):
    """state_dict order: per block scale_net.{0,2,..}.{weight,bias} then translate_net."""
    return np.concatenate([v.detach().cpu().numpy().astype(np.float32).ravel()
                           for v in netG.state_dict().values()])


def pack_grads(netG):
    return np.concatenate([p.grad.detach().cpu().numpy().astype(np.float32).ravel()
                           for p in netG.parameters()])


def pack_adam(optimizer, netG, key):
    return np.concatenate([optimizer.state[p][key].detach().cpu().numpy().astype(np.float32).ravel()
                           for p in netG.parameters()])


def make_trainer(D, H=16, B=3, L=1, seed=0, lr=1e-3):
    torch.manual_seed(seed)
    return Trainer(D, hidden_dim=H, num_blocks=B, num_layers=L, flow='nvp', log_dir=None,
                   learning_rate=lr, log_level=logging.WARNING)


# ----------------------------------------------------------------------------------------------
# G1
# ----------------------------------------------------------------------------------------------
def gen_flow():
    cases = [  # (name, D, H, B, L)
        ('d2', 2, 16, 3, 1), ('d3', 3, 16, 3, 1), ('d5', 5, 16, 3, 1), ('d20', 20, 16, 3, 1),
        ('d32', 32, 16, 3, 1), ('d50', 50, 16, 3, 1), ('d100', 100, 16, 3, 1),
        ('d7_l2', 7, 16, 3, 2), ('d6_l0', 6, 16, 2, 0), ('d4_h32_b5', 4, 32, 5, 1),
        # round-5 verdict item 8: shapes the reference's Trainer accepts that are not multiples of the MFMA tile (hidden_dim padded
        # to the next 16 with zero weights: exact), and more hidden layers
        ('d5_h10', 5, 10, 3, 1), ('d6_h24_l3', 6, 24, 2, 3), ('d9_h40', 9, 40, 3, 1),
    ]
    for name, D, H, B, L in cases:
        if not wanted(name):
            continue
        np.random.seed(0)
        t = make_trainer(D, H, B, L, seed=0)
        x = np.random.uniform(-1, 1, size=(64, D))
        out = {'D': D, 'H': H, 'B': B, 'L': L, 'x': x}
        for tag in ('init', 'trained'):
            if tag == 'trained':
                np.random.seed(1)
                torch.manual_seed(1)
                live = np.random.uniform(-1, 1, size=(300, D))
                t.train(live, max_iters=20, jitter=0.01)
            z, ldf = t.forward(x, to_numpy=True)
            xb, ldi = t.inverse(z, to_numpy=True)
            lp = t.log_probs(x, to_numpy=True)
            out.update({'w_' + tag: pack_state_dict(t.netG), 'z_' + tag: z, 'ldf_' + tag: ldf,
                        'xb_' + tag: xb, 'ldi_' + tag: ldi, 'lp_' + tag: lp})
        # the reference's own round-trip bound (tests/test_flows.py:8, :27-30)
        assert np.abs(np.max(out['xb_trained'] - x.astype(np.float32))) <= 1e-5
        np.savez_compressed(os.path.join(OUT, 'flow_%s.npz' % name), **out)
        print('G1 flow', name, 'nparams', out['w_init'].size)


# ----------------------------------------------------------------------------------------------
# G2 / G3
# ----------------------------------------------------------------------------------------------
def gen_like():
    specs = [('rosenbrock', Rosenbrock, 5.0, [2, 3, 50, 100]),
             ('gaussmix', GaussianMix, 10.0, [2, 20]),
             ('himmelblau', Himmelblau, 5.0, [2])]
    out = {}
    for name, cls, scale, dims in specs:
        for D in dims:
            np.random.seed(100 + D)
            like = cls(D)
            x64 = np.random.uniform(-1, 1, size=(64, D))
            x32 = x64.astype(np.float32)
            tmp = tempfile.mkdtemp()
            s = NestedSampler(D, like, transform=lambda x, s=scale: s * x, log_dir=tmp, flow='nvp',
                              num_live_points=10, log_level=logging.WARNING)
            l64, _ = s.loglike(x64)
            l32, _ = s.loglike(x32)
            shutil.rmtree(tmp, ignore_errors=True)
            key = '%s_d%d' % (name, D)
            out[key + '_x64'] = x64
            out[key + '_l64'] = np.asarray(l64, dtype=np.float64)
            out[key + '_l32'] = np.asarray(l32, dtype=np.float64)  # value computed from fp32 inputs
            out[key + '_l32_dtype'] = str(np.asarray(l32).dtype)
            out[key + '_scale'] = scale
            print('G2', key, np.asarray(l32).dtype, l64[:2], l32[:2])
    np.savez_compressed(os.path.join(OUT, 'like.npz'), **out)

    # G3: rows straddling the box edge
    D = 6
    np.random.seed(7)
    x = np.random.uniform(-1.05, 1.05, size=(200, D))
    x[0] = 1.0
    x[1] = -1.0
    x[2] = np.nextafter(np.float32(1.0), np.float32(2.0))
    x[3, 2] = np.nan
    tmp = tempfile.mkdtemp()
    s = NestedSampler(D, Rosenbrock(D), transform=lambda x: 5 * x, log_dir=tmp, flow='nvp',
                      num_live_points=10, log_level=logging.WARNING)
    flags64 = s.prior(x)
    flags32 = s.prior(x.astype(np.float32))
    shutil.rmtree(tmp, ignore_errors=True)
    np.savez_compressed(os.path.join(OUT, 'prior.npz'), x=x, flag64=flags64, flag32=flags32)
    print('G3 prior in-box', int(np.sum(flags64 == 0)), 'of', len(x))


def gen_like2():
    """G2b: the remaining analytic likelihoods (nnest/likelihoods.py:77-150) through safe_loglike"""
    specs = [('gaussian', lambda D: Gaussian(D, 0.99, lim=3), 3.0, [2, 5, 20], (0.99,)),
             ('gaussian_c0', lambda D: Gaussian(D, 0.3), 3.0, [4], (0.3,)),
             ('eggbox', lambda D: Eggbox(D), float(5 * np.pi), [2], ()),
             ('shell', lambda D: GaussianShell(D), 5.0, [2, 10], (0.1, 2, 0)),
             ('shell_c', lambda D: GaussianShell(D, sigma=0.3, rshell=1.5, center=1), 5.0, [3], (0.3, 1.5, 1)),
             ('double_shell', lambda D: DoubleGaussianShell(D), 6.0, [2, 5], (0.1, 2, -4, 0.1, 2, 4))]
    out = {}
    for name, mk, scale, dims, params in specs:
        for D in dims:
            np.random.seed(200 + D)
            like = mk(D)
            x64 = np.random.uniform(-1, 1, size=(64, D))
            x32 = x64.astype(np.float32)
            tmp = tempfile.mkdtemp()
            s = NestedSampler(D, like, transform=lambda x, s=scale: s * x, log_dir=tmp, flow='nvp',
                              num_live_points=10, log_level=logging.WARNING)
            l64, _ = s.loglike(x64)
            l32, _ = s.loglike(x32)
            shutil.rmtree(tmp, ignore_errors=True)
            key = '%s_d%d' % (name, D)
            out[key + '_x64'] = x64
            out[key + '_l64'] = np.asarray(l64, dtype=np.float64)
            out[key + '_l32'] = np.asarray(l32, dtype=np.float64)
            out[key + '_scale'] = scale
            out[key + '_params'] = np.asarray(params, dtype=np.float64)
            print('G2b', key, np.asarray(l32).dtype, l64[:2], l32[:2])
    np.savez_compressed(os.path.join(OUT, 'like2.npz'), **out)


# ----------------------------------------------------------------------------------------------
# G4: minibatch steps of Trainer._train with recorded shuffle and jitter noise
# ----------------------------------------------------------------------------------------------
def replay_loader_rng(n, batch, D):
    """Re-draw what DataLoader(shuffle=True) + `torch.randn_like(data)` (trainer.py:185, :392)
    consume from the global torch generator for ONE epoch, starting from its current state."""
    _base_seed = torch.empty((), dtype=torch.int64).random_().item()  # _BaseDataLoaderIter
    seed = int(torch.empty((), dtype=torch.int64).random_().item())  # RandomSampler.__iter__
    g = torch.Generator()
    g.manual_seed(seed)
    perm = torch.randperm(n, generator=g)
    noises = []
    for b in range(0, n, batch):
        m = min(batch, n - b)
        noises.append(torch.randn(m, D))
    return perm.numpy().astype(np.int32), torch.cat(noises, 0).numpy()


def gen_train_steps():
    for name, D, n, L, H in [('d50', 50, 250, 1, 16), ('d5', 5, 230, 1, 16), ('d7_l2', 7, 120, 2, 16),
                             ('d5_h10', 5, 230, 1, 10), ('d6_l3', 6, 120, 3, 16), ('d8_h24_l2', 8, 120, 2, 24)]:
        if not wanted(name):
            continue
        np.random.seed(3)
        t = make_trainer(D, H, 3, L, seed=0)
        X = np.random.uniform(-1, 1, size=(n, D)).astype(np.float32)
        jitter = 0.02
        w0 = pack_state_dict(t.netG)
        # (a) explicit loop with the reference's model + optimizer, capturing every gradient
        t2 = copy.deepcopy(t)
        t2.optimizer = torch.optim.Adam(t2.netG.parameters(), lr=1e-3, weight_decay=1e-6)
        n_epochs = 2
        torch.manual_seed(11)
        rng_state = torch.get_rng_state()
        perms, noises = [], []
        for e in range(n_epochs):
            p, nz = replay_loader_rng(n, 100, D)
            perms.append(p)
            noises.append(nz)
        losses, grads, ws, ms, vs = [], [], [], [], []
        t2.netG.train()
        Xt = torch.from_numpy(X)
        for e in range(n_epochs):
            for b in range(0, n, 100):
                idx = torch.from_numpy(perms[e][b:b + 100].astype(np.int64))
                data = Xt[idx] + jitter * torch.from_numpy(noises[e][b:b + 100])
                t2.optimizer.zero_grad()
                loss = -t2.netG.log_probs(data).mean()
                loss.backward()
                grads.append(pack_grads(t2.netG))
                t2.optimizer.step()
                losses.append(loss.item())
                ws.append(pack_state_dict(t2.netG))
                ms.append(pack_adam(t2.optimizer, t2.netG, 'exp_avg'))
                vs.append(pack_adam(t2.optimizer, t2.netG, 'exp_avg_sq'))
        # (b) the reference's own _train on the same generator state must land on the same weights
        torch.set_rng_state(rng_state)
        ds = torch.utils.data.TensorDataset(Xt)
        loader = torch.utils.data.DataLoader(ds, batch_size=100, shuffle=True)
        ref_epoch_losses = [t._train(e + 1, loader, jitter=jitter) for e in range(n_epochs)]
        assert np.array_equal(pack_state_dict(t.netG), ws[-1]), 'replay differs from reference _train'
        # validation pass (trainer.py:405-418): one full batch, mean / len
        vloader = torch.utils.data.DataLoader(ds, batch_size=n, shuffle=False, drop_last=True)
        vloss = t._validate(1, vloader)
        np.savez_compressed(
            os.path.join(OUT, 'train_%s.npz' % name), D=D, H=H, B=3, L=L, X=X, jitter=jitter, batch=100,
            lr=1e-3, weight_decay=1e-6, w0=w0, perms=np.stack(perms), noises=np.stack(noises),
            losses=np.array(losses), grads=np.stack(grads), ws=np.stack(ws), ms=np.stack(ms),
            vs=np.stack(vs), ref_epoch_losses=np.array(ref_epoch_losses), valid_loss=vloss)
        print('G4 train', name, 'steps', len(losses), 'loss0', losses[0], 'ref epoch loss', ref_epoch_losses)


# ----------------------------------------------------------------------------------------------
# G7: Trainer.train() (split + epochs + early stopping + best restore)
# ----------------------------------------------------------------------------------------------
def gen_train_run():
    for name, D, N, iters, patience, batch in [('d5', 5, 200, 6, 50, 100), ('d20_pat', 20, 300, 40, 3, 100),
                                               ('d20_b256', 20, 700, 6, 50, 256)]:
        if not wanted(name):
            continue
        np.random.seed(5)
        t = make_trainer(D, 16, 3, 1, seed=2)
        t.batch_size = batch   # Trainer(batch_size=...)  trainer.py:76
        live = np.random.uniform(-1, 1, size=(N, D))
        w0 = pack_state_dict(t.netG)
        # what train() will consume from the numpy global RNG: sklearn ShuffleSplit -> rng.permutation(N)
        np_state = np.random.get_state()
        perm_split = np.random.permutation(N)
        np.random.set_state(np_state)
        n_valid = int(np.ceil(0.1 * N))
        n_train = N - n_valid
        torch.manual_seed(21)
        rng_state = torch.get_rng_state()
        perms, noises = [], []
        for e in range(iters):
            p, nz = replay_loader_rng(n_train, batch, D)
            perms.append(p)
            noises.append(nz)
            # Trainer._validate iterates the validation loader once per epoch (trainer.py:203, :410): every DataLoader iterator draws
            # its _base_seed from the global generator, shuffled or not.  (Rounds 1-5 left this draw out: the recorded perms / noises
            # of epochs >= 2 were not the ones the reference consumed -- hidden by the 4-decimal logged losses; the replay is now
            # asserted against the reference's final weights below.)
            torch.empty((), dtype=torch.int64).random_()
        torch.set_rng_state(rng_state)
        t_replay = copy.deepcopy(t)
        t_replay.optimizer = torch.optim.Adam(t_replay.netG.parameters(), lr=1e-3, weight_decay=1e-6)
        # capture per-epoch losses through the reference logger-free route: wrap nothing, just
        # call train() and afterwards recompute the curve from recorded weights is not possible,
        # so record them by calling the reference's _train/_validate through train() with
        # log_interval=1 and a logging handler.
        recs = []

        class H(logging.Handler):
            def emit(self, record):
                recs.append(record.getMessage())

        h = H()
        t.logger.addHandler(h)
        t.logger.setLevel(logging.INFO)
        jitter = 0.01
        t.train(live, max_iters=iters, jitter=jitter, log_interval=1, patience=patience)
        t.logger.removeHandler(h)
        ep = [r for r in recs if r.startswith('Epoch [') and 'train loss' in r]
        tl = np.array([float(r.split('train loss [')[1].split(']')[0]) for r in ep])
        vl = np.array([float(r.split('validation loss [')[1].split(']')[0]) for r in ep])
        # confirm the split replay against what sklearn actually produced
        from sklearn.model_selection import train_test_split
        np.random.set_state(np_state)
        Xtr, Xva = train_test_split(live, test_size=0.1)
        assert np.array_equal(Xva, live[perm_split[:n_valid]])
        assert np.array_equal(Xtr, live[perm_split[n_valid:n_valid + n_train]])
        # confirm the loader replay: the reference's model stepped over the recorded minibatches lands on the weights train() left
        # (when patience did not stop the run early: train() then restores an earlier epoch's weights)
        if len(ep) == iters and t.best_validation_epoch == iters:
            Xt_ = torch.from_numpy(Xtr.astype(np.float32))
            t_replay.netG.train()
            for e in range(iters):
                for b in range(0, n_train, batch):
                    idx = torch.from_numpy(perms[e][b:b + batch].astype(np.int64))
                    data = Xt_[idx] + jitter * torch.from_numpy(noises[e][b:b + batch])
                    t_replay.optimizer.zero_grad()
                    (-t_replay.netG.log_probs(data).mean()).backward()
                    t_replay.optimizer.step()
            assert np.array_equal(pack_state_dict(t_replay.netG), pack_state_dict(t.netG)), 'loader replay differs from train()'
            print('G7 trainrun', name, 'replay == train() bit for bit')
        np.savez_compressed(
            os.path.join(OUT, 'trainrun_%s.npz' % name), D=D, H=16, B=3, L=1, live=live, w0=w0,
            jitter=jitter, batch=batch, lr=1e-3, weight_decay=1e-6, patience=patience, max_iters=iters,
            perm_split=perm_split.astype(np.int32), perms=np.stack(perms), noises=np.stack(noises),
            train_losses_logged=tl, valid_losses_logged=vl, w_final=pack_state_dict(t.netG),
            best_validation_loss=t.best_validation_loss, best_validation_epoch=t.best_validation_epoch,
            epochs_run=len(ep))
        print('G7 trainrun', name, 'epochs', len(ep), 'best', t.best_validation_epoch, t.best_validation_loss)


# ----------------------------------------------------------------------------------------------
# G5: _mcmc_sample traces
# ----------------------------------------------------------------------------------------------
def gen_mcmc():
    cases = [('rosen_d2', Rosenbrock, 2, 5.0, 16, 24, False), ('rosen_d2_dyn', Rosenbrock, 2, 5.0, 10, 40, True),
             ('rosen_d50', Rosenbrock, 50, 5.0, 16, 12, False), ('rosen_d50_dyn', Rosenbrock, 50, 5.0, 16, 30, True),
             ('gmix_d20', GaussianMix, 20, 10.0, 16, 12, False), ('himmel_d2', Himmelblau, 2, 5.0, 16, 16, True),
             # loglstar=None: likelihood and prior in the proposal ratio (sampler.py:371-410); stored as loglstar = NaN
             ('free_rosen_d2', Rosenbrock, 2, 5.0, 16, 40, True), ('free_gmix_d20', GaussianMix, 20, 10.0, 16, 20, False)]
    for name, cls, D, scale, C, S, dyn in cases:
        np.random.seed(9)
        torch.manual_seed(9)
        like = cls(D)
        tmp = tempfile.mkdtemp()
        s = NestedSampler(D, like, transform=lambda x, sc=scale: sc * x, log_dir=tmp, flow='nvp',
                          num_live_points=400, learning_rate=1e-3, log_level=logging.WARNING)
        live_u = s.sample_prior(400)
        live_logl, _ = s.loglike(live_u)
        # shrink towards higher likelihood so that constrained moves are sometimes accepted
        order = np.argsort(live_logl)
        live_u = live_u[order[100:]]
        live_logl = live_logl[order[100:]]
        s.trainer.train(live_u, max_iters=15, jitter=0.01)
        s.trainer.path = None
        loglstar = None if name.startswith('free_') else float(np.min(live_logl))
        idx = np.random.randint(0, live_u.shape[0], size=C)
        init = live_u[idx]
        init_l = live_logl[idx]
        step = (0.2 if name.startswith('free_') else 1.0) / np.sqrt(D)
        torch.manual_seed(77)
        rng_state = torch.get_rng_state()
        dz = np.stack([torch.randn(C, D).numpy() for _ in range(1)])  # placeholder to learn the order
        torch.set_rng_state(rng_state)
        dzs, us = [], []
        for _ in range(S):
            dzs.append(torch.randn(C, D).numpy())   # torch.randn_like(z)   sampler.py:310
            us.append(torch.rand(C).numpy())        # torch.rand(shape)     sampler.py:334
        torch.set_rng_state(rng_state)
        calls0 = s.total_calls
        samples, latent, derived, loglikes, scale_out, ncall = s._mcmc_sample(
            S, init_samples=init, init_loglikes=init_l, init_derived=np.empty((C, 0)),
            loglstar=loglstar, step_size=step, dynamic_step_size=dyn, plot_trace=False)
        np.savez_compressed(
            os.path.join(OUT, 'mcmc_%s.npz' % name), D=D, H=16, B=3, L=1, like=cls.__name__, scale=scale,
            w=pack_state_dict(s.trainer.netG), init=init, init_logl=init_l, loglstar=np.nan if loglstar is None else loglstar, step=step,
            dynamic=dyn, dz=np.stack(dzs), u=np.stack(us), samples=samples, latent=latent,
            loglikes=loglikes, scale_out=scale_out, ncall=ncall, total_calls=s.total_calls - calls0,
            total_accepted=int(s.total_accepted), total_rejected=int(s.total_rejected))
        shutil.rmtree(tmp, ignore_errors=True)
        moved = np.mean(np.any(samples[:, 0] != samples[:, -1], axis=1))
        print('G5 mcmc', name, 'ncall', ncall, 'scale', scale_out, 'moved frac', moved,
              'acc', s.total_accepted, 'rej', s.total_rejected)


def gen_mcmc_spline():
    """G5b mcmc_spline_*.npz: Sampler._mcmc_sample (nnest/sampler.py:291-444) on the reference's DEFAULT flow (flow='spline':
    NSF_CL couplings, nnest/networks.py:458-556, behind ActNorm + 1x1 conv) with the torch.randn_like / torch.rand draws recorded:
    every accept / reject decision of the reference itself, fixed and dynamic step, x_dim 5 and 50 (round-5 verdict item 2)."""
    cases = [('rosen_d5', Rosenbrock, 5, 5.0, 16, 24, False, 10, 1.0), ('rosen_d5_dyn', Rosenbrock, 5, 5.0, 16, 40, True, 10, 1.0),
             ('rosen_d50', Rosenbrock, 50, 5.0, 16, 16, False, 6, 0.25), ('rosen_d50_dyn', Rosenbrock, 50, 5.0, 16, 30, True, 6, 1.0),
             ('rosen_d50_c40_dyn', Rosenbrock, 50, 5.0, 40, 16, True, 6, 1.0)]
    for name, cls, D, scale, C, S, dyn, iters, step_f in cases:
        np.random.seed(9)
        torch.manual_seed(9)
        like = cls(D)
        tmp = tempfile.mkdtemp()
        s = NestedSampler(D, like, transform=lambda x, sc=scale: sc * x, log_dir=tmp, flow='spline',
                          num_live_points=400, learning_rate=1e-3, log_level=logging.WARNING)
        live_u = s.sample_prior(400)
        live_logl, _ = s.loglike(live_u)
        order = np.argsort(live_logl)
        live_u = live_u[order[100:]]
        live_logl = live_logl[order[100:]]
        s.trainer.train(live_u, max_iters=iters, jitter=0.01)
        s.trainer.path = None
        loglstar = float(np.min(live_logl))
        idx = np.random.randint(0, live_u.shape[0], size=C)
        init = live_u[idx]
        init_l = live_logl[idx]
        step = step_f / np.sqrt(D)
        torch.manual_seed(77)
        rng_state = torch.get_rng_state()
        dzs, us = [], []
        for _ in range(S):
            dzs.append(torch.randn(C, D).numpy())   # torch.randn_like(z)   sampler.py:310
            us.append(torch.rand(C).numpy())        # torch.rand(shape)     sampler.py:334
        torch.set_rng_state(rng_state)
        calls0, acc0, rej0 = s.total_calls, s.total_accepted, s.total_rejected
        samples, latent, derived, loglikes, scale_out, ncall = s._mcmc_sample(
            S, init_samples=init, init_loglikes=init_l, init_derived=np.empty((C, 0)),
            loglstar=loglstar, step_size=step, dynamic_step_size=dyn, plot_trace=False)
        np.savez_compressed(
            os.path.join(OUT, 'mcmc_spline_%s.npz' % name), D=D, H=16, B=3, K=8, tail=3.0, like=cls.__name__, scale=scale,
            w=pack_state_dict(s.trainer.netG), P=spline_P(s.trainer.netG), init=init, init_logl=init_l, loglstar=loglstar, step=step,
            dynamic=dyn, dz=np.stack(dzs), u=np.stack(us), samples=samples, latent=latent,
            loglikes=loglikes, scale_out=scale_out, ncall=ncall, total_calls=s.total_calls - calls0,
            total_accepted=int(s.total_accepted - acc0), total_rejected=int(s.total_rejected - rej0))
        shutil.rmtree(tmp, ignore_errors=True)
        moved = np.mean(np.any(samples[:, 0] != samples[:, -1], axis=1))
        print('G5b mcmc spline', name, 'ncall', ncall, 'scale', scale_out, 'moved frac', moved,
              'acc', s.total_accepted - acc0, 'rej', s.total_rejected - rej0)


# ----------------------------------------------------------------------------------------------
# G6: seeded config-1 end-to-end
# ----------------------------------------------------------------------------------------------
def gen_nested():
    np.random.seed(0)
    torch.manual_seed(0)
    tmp = tempfile.mkdtemp()
    like = Rosenbrock(2)
    s = NestedSampler(2, like, transform=lambda x: 5 * x, log_dir=tmp, num_live_points=100,
                      hidden_dim=16, num_layers=1, num_blocks=3, flow='nvp', log_level=logging.WARNING)
    s.run(train_iters=2000, mcmc_num_chains=10)
    import csv
    with open(os.path.join(s.logs['results'], 'final.csv')) as f:
        rows = list(csv.reader(f))
    res = dict(zip(rows[0], [float(v) for v in rows[1]]))
    res['config'] = 'Rosenbrock x_dim=2, 100 live points, nvp h16 b3 l1, train_iters=2000, mcmc_num_chains=10, seeds 0/0'
    res['nsamples'] = int(s.samples.shape[0])
    res['posterior_mean'] = (np.sum(s.samples * s.weights[:, None], 0) / np.sum(s.weights)).tolist()
    with open(os.path.join(OUT, 'nested_cfg1.json'), 'w') as f:
        json.dump(res, f, indent=1)
    shutil.rmtree(tmp, ignore_errors=True)
    print('G6', res)


# ----------------------------------------------------------------------------------------------
# G14: what the reference writes to disk during the seeded config-1 run (nnest/sampler.py:494-511; nnest/nested.py:92-95,
# :473-485, :503-506): the text products as they are (first / last lines) and one complete checkpoint set as arrays, so that
# tests can check format equality and resume a run of this build from a reference-written checkpoint.
# ----------------------------------------------------------------------------------------------
def gen_formats():
    np.random.seed(0)
    torch.manual_seed(0)
    tmp = tempfile.mkdtemp()
    s = NestedSampler(2, Rosenbrock(2), transform=lambda x: 5 * x, log_dir=tmp, num_live_points=100,
                      hidden_dim=16, num_layers=1, num_blocks=3, flow='nvp', log_level=logging.WARNING)
    s.run(train_iters=2000, mcmc_num_chains=10)
    out = os.path.join(OUT, 'formats')
    os.makedirs(out, exist_ok=True)

    def head_tail(src, dst, nh, nt):
        with open(src) as f:
            lines = f.read().split('\n')
        body = [ln for ln in lines if ln != '']
        with open(dst, 'w') as f:
            f.write('\n'.join(body[:nh] + (['...'] if len(body) > nh + nt else []) + (body[-nt:] if nt else [])) + '\n')
        return len(body)

    n_chain = head_tail(os.path.join(s.logs['chains'], 'chain.txt'), os.path.join(out, 'chain_head_tail.txt'), 6, 3)
    head_tail(os.path.join(s.logs['results'], 'results.csv'), os.path.join(out, 'results_head.csv'), 3, 0)
    shutil.copy(os.path.join(s.logs['results'], 'final.csv'), os.path.join(out, 'final.csv'))
    with open(os.path.join(s.logs['info'], 'params.txt')) as f:
        params = json.load(f)
    cps = sorted(int(f.split('_')[1].split('.')[0]) for f in os.listdir(s.logs['checkpoint']) if f.startswith('checkpoint_'))
    # the LAST checkpoint is the one consistent with saved_*.npy (those are overwritten at every checkpoint)
    it = cps[-1]
    cp = s.logs['checkpoint']
    with open(os.path.join(cp, 'checkpoint_%d.txt' % it)) as f:
        state = json.load(f)
    np.savez_compressed(os.path.join(out, 'checkpoint_set.npz'), it=it,
                        active_u=np.load(os.path.join(cp, 'active_u_%d.npy' % it)),
                        active_v=np.load(os.path.join(cp, 'active_v_%d.npy' % it)),
                        active_logl=np.load(os.path.join(cp, 'active_logl_%d.npy' % it)),
                        active_derived=np.load(os.path.join(cp, 'active_derived_%d.npy' % it)),
                        saved_v=np.load(os.path.join(cp, 'saved_v.npy')), saved_logl=np.load(os.path.join(cp, 'saved_logl.npy')),
                        saved_logwt=np.load(os.path.join(cp, 'saved_logwt.npy')))
    meta = dict(checkpoint_iterations=cps, checkpoint_state=state, params_keys=sorted(params.keys()), chain_rows=n_chain,
                run_dir_layout=sorted(os.listdir(s.logs['run_dir'])), checkpoint_files=sorted(os.listdir(cp))[:12],
                final_logz=float(s.logz), config='Rosenbrock x_dim=2, 100 live points, nvp h16 b3 l1, train_iters=2000, '
                                                 'mcmc_num_chains=10, seeds 0/0 (the G6 run)')
    with open(os.path.join(out, 'meta.json'), 'w') as f:
        json.dump(meta, f, indent=1)
    shutil.rmtree(tmp, ignore_errors=True)
    print('G14 formats', meta['checkpoint_iterations'][-3:], n_chain)


# ----------------------------------------------------------------------------------------------
# G8: SingleSpeedNVP scale variants (networks.py:328-347): scale='translate' (translate-only couplings) and
# scale='constant' (translate-only couplings + one ScaleLayer scalar after each): passes, a trained state, and
# minibatch steps with every gradient.  Vectors are the concatenated state_dict (no scale nets in these variants).
# ----------------------------------------------------------------------------------------------
def gen_scale_variants():
    for scale in ('translate', 'constant'):
        for name, D, L in [('d5', 5, 1), ('d50', 50, 1), ('d8_l2', 8, 2)]:
            np.random.seed(0)
            torch.manual_seed(4)
            t = Trainer(D, hidden_dim=16, num_blocks=3, num_layers=L, flow='nvp', log_dir=None, scale=scale,
                        learning_rate=1e-3, log_level=logging.WARNING)
            if scale == 'constant':   # ScaleLayer scalars start at 0 (networks.py:316): move them off zero
                with torch.no_grad():
                    for k, v in t.netG.state_dict().items():
                        if k.endswith('.scale'):
                            v.fill_(float(np.random.uniform(-0.3, 0.3)))
            x = np.random.uniform(-1, 1, size=(64, D))
            out = {'D': D, 'H': 16, 'B': 3, 'L': L, 'x': x, 'keys': np.array(list(t.netG.state_dict().keys()))}
            w0 = pack_state_dict(t.netG)
            # minibatch steps (as G4) from the initial state
            n, jitter = 230, 0.02
            X = np.random.uniform(-1, 1, size=(n, D)).astype(np.float32)
            t2 = copy.deepcopy(t)
            t2.optimizer = torch.optim.Adam(t2.netG.parameters(), lr=1e-3, weight_decay=1e-6)
            torch.manual_seed(11)
            rng_state = torch.get_rng_state()
            perms, noises = [], []
            for e in range(2):
                pp, nz = replay_loader_rng(n, 100, D)
                perms.append(pp)
                noises.append(nz)
            losses, grads, ws = [], [], []
            t2.netG.train()
            Xt = torch.from_numpy(X)
            for e in range(2):
                for b in range(0, n, 100):
                    idx = torch.from_numpy(perms[e][b:b + 100].astype(np.int64))
                    data = Xt[idx] + jitter * torch.from_numpy(noises[e][b:b + 100])
                    t2.optimizer.zero_grad()
                    loss = -t2.netG.log_probs(data).mean()
                    loss.backward()
                    grads.append(pack_grads(t2.netG))
                    t2.optimizer.step()
                    losses.append(loss.item())
                    ws.append(pack_state_dict(t2.netG))
            # pass outputs at the initial state, before the reference's own _train moves `t`
            z, ldf = t.forward(x, to_numpy=True)
            xb, ldi = t.inverse(z, to_numpy=True)
            out.update(w_init=w0, z_init=z, ldf_init=ldf, xb_init=xb, ldi_init=ldi, lp_init=t.log_probs(x, to_numpy=True))
            tt = t
            torch.set_rng_state(rng_state)
            ds = torch.utils.data.TensorDataset(Xt)
            loader = torch.utils.data.DataLoader(ds, batch_size=100, shuffle=True)
            for e in range(2):
                tt._train(e + 1, loader, jitter=jitter)
            assert np.array_equal(pack_state_dict(tt.netG), ws[-1]), 'replay differs from reference _train'
            out.update(X=X, jitter=jitter, w0=w0, perms=np.stack(perms), noises=np.stack(noises),
                       losses=np.array(losses), grads=np.stack(grads), ws=np.stack(ws))
            for tag in ('trained',):
                np.random.seed(1)
                torch.manual_seed(1)
                t.train(np.random.uniform(-1, 1, size=(300, D)), max_iters=20, jitter=0.01)
                z, ldf = t.forward(x, to_numpy=True)
                xb, ldi = t.inverse(z, to_numpy=True)
                lp = t.log_probs(x, to_numpy=True)
                out.update({'w_' + tag: pack_state_dict(t.netG), 'z_' + tag: z, 'ldf_' + tag: ldf,
                            'xb_' + tag: xb, 'ldi_' + tag: ldi, 'lp_' + tag: lp})
            np.savez_compressed(os.path.join(OUT, 'scale_%s_%s.npz' % (scale, name)), **out)
            print('G8 scale', scale, name, 'nparams', w0.size, 'loss0', losses[0])


# ----------------------------------------------------------------------------------------------
# G9: neural-spline flow, SingleSpeedSpline (networks.py:393-715): [ActNorm, Invertible1x1Conv, NSF_CL] x B.
# The fixed permutations P of the 1x1 convs are plain attributes (not in the state_dict): stored beside the weights.
# Sequence recorded (ActNorm initialises itself from the first batch that is pushed FORWARD, networks.py:698-705):
#   raw state -> inverse(z0) -> forward(x_first) [data-dependent init happens here] -> forward / inverse / log_probs
#   on a second batch -> one loss + autograd gradient -> Adam minibatch steps -> a trained state.
# ----------------------------------------------------------------------------------------------
def spline_P(netG):
    return np.stack([f.P.detach().cpu().numpy().astype(np.float32) for f in netG.flow.flows if hasattr(f, 'P')])


def gen_spline():
    for name, D, H, B in [('d2', 2, 16, 3), ('d5', 5, 16, 3), ('d8_h32', 8, 32, 2), ('d50', 50, 16, 3), ('d6_h10', 6, 10, 3)]:
        if not wanted(name):
            continue
        np.random.seed(0)
        torch.manual_seed(6)
        t = Trainer(D, hidden_dim=H, num_blocks=B, flow='spline', log_dir=None, learning_rate=1e-3,
                    log_level=logging.WARNING)
        out = {'D': D, 'H': H, 'B': B, 'K': 8, 'tail': 3.0, 'P': spline_P(t.netG),
               'keys': np.array(list(t.netG.state_dict().keys()))}
        out['w_raw'] = pack_state_dict(t.netG)
        z0 = (0.8 * np.random.randn(48, D)).astype(np.float32)
        z0[:3] *= 5.0                                   # rows reaching outside the spline interval [-3, 3]
        xi, ldi = t.inverse(z0, to_numpy=True)
        out.update(z0=z0, x_inv_raw=xi, ld_inv_raw=ldi)
        x_first = np.random.uniform(-1, 1, size=(100, D)).astype(np.float32)
        zf, ldf = t.forward(x_first, to_numpy=True)     # ActNorm data-dependent init
        out.update(x_first=x_first, z_first=zf, ld_first=ldf, w_init=pack_state_dict(t.netG))
        x = np.random.uniform(-1, 1, size=(64, D)).astype(np.float32)
        x[:2] *= 6.0
        # gradient + Adam steps from the initialised state
        n, jitter = 230, 0.02
        X = np.random.uniform(-1, 1, size=(n, D)).astype(np.float32)
        t2 = copy.deepcopy(t)
        t2.optimizer = torch.optim.Adam(t2.netG.parameters(), lr=1e-3, weight_decay=1e-6)
        torch.manual_seed(11)
        rng_state = torch.get_rng_state()
        perms, noises = [], []
        for e in range(2):
            pp, nz = replay_loader_rng(n, 100, D)
            perms.append(pp)
            noises.append(nz)
        losses, grads, ws = [], [], []
        Xt = torch.from_numpy(X)
        t2.netG.train()
        for e in range(2):
            for b in range(0, n, 100):
                idx = torch.from_numpy(perms[e][b:b + 100].astype(np.int64))
                data = Xt[idx] + jitter * torch.from_numpy(noises[e][b:b + 100])
                t2.optimizer.zero_grad()
                loss = -t2.netG.log_probs(data).mean()
                loss.backward()
                grads.append(pack_grads(t2.netG))
                t2.optimizer.step()
                losses.append(loss.item())
                ws.append(pack_state_dict(t2.netG))
        out.update(X=X, jitter=jitter, perms=np.stack(perms), noises=np.stack(noises), losses=np.array(losses),
                   grads=np.stack(grads[:2]), ws=np.stack([ws[0], ws[-1]]))
        for tag in ('init', 'trained'):
            if tag == 'trained':
                # the reference's own _train on the same generator state must land on the replayed weights
                torch.set_rng_state(rng_state)
                ds = torch.utils.data.TensorDataset(Xt)
                loader = torch.utils.data.DataLoader(ds, batch_size=100, shuffle=True)
                for e in range(2):
                    t._train(e + 1, loader, jitter=jitter)
                assert np.array_equal(pack_state_dict(t.netG), ws[-1]), 'replay differs from reference _train'
                np.random.seed(1)
                torch.manual_seed(1)
                t.train(np.random.uniform(-1, 1, size=(300, D)), max_iters=30, jitter=0.01)
            z, ldz = t.forward(x, to_numpy=True)
            xb, ldb = t.inverse(z, to_numpy=True)
            lp = t.log_probs(x, to_numpy=True)
            zs = (0.9 * np.random.RandomState(3).randn(64, D)).astype(np.float32)
            xs, lds = t.inverse(zs, to_numpy=True)
            out.update({'w_' + tag: pack_state_dict(t.netG), 'x': x, 'z_' + tag: z, 'ldf_' + tag: ldz, 'xb_' + tag: xb,
                        'ldi_' + tag: ldb, 'lp_' + tag: lp, 'zs': zs, 'xs_' + tag: xs, 'lds_' + tag: lds})
        np.savez_compressed(os.path.join(OUT, 'spline_%s.npz' % name), **out)
        print('G9 spline', name, 'nparams', out['w_raw'].size, 'loss0', losses[0], 'roundtrip',
              np.abs(out['xb_trained'] - x).max())


# ----------------------------------------------------------------------------------------------
# G10: generalised-normal base distribution (nnest/distributions/generalised_normal.py; examples/nested/run.py:20-21,
# --base_dist gen_normal --beta 8): log_probs, the loss gradient and Adam steps for both flows
# ----------------------------------------------------------------------------------------------
def gen_base_dist():
    from nnest.distributions.generalised_normal import GeneralisedNormal
    for flow, D in (('nvp', 5), ('spline', 5), ('nvp', 50)):
        np.random.seed(0)
        torch.manual_seed(8)
        beta = 8.0
        base = GeneralisedNormal(torch.zeros(D), torch.ones(D), torch.tensor(beta))
        t = Trainer(D, hidden_dim=16, num_blocks=3, num_layers=1, flow=flow, log_dir=None, learning_rate=1e-3,
                    base_dist=base, log_level=logging.WARNING)
        out = {'D': D, 'beta': beta, 'flow': flow}
        if flow == 'spline':
            out['P'] = spline_P(t.netG)
            x_first = np.random.uniform(-1, 1, size=(100, D)).astype(np.float32)
            t.forward(x_first)           # ActNorm data-dependent init
        out['w0'] = pack_state_dict(t.netG)
        x = np.random.uniform(-1, 1, size=(64, D)).astype(np.float32)
        out['x'] = x
        out['lp0'] = t.log_probs(x, to_numpy=True)
        n, jitter = 230, 0.02
        X = np.random.uniform(-1, 1, size=(n, D)).astype(np.float32)
        torch.manual_seed(11)
        perms, noises = [], []
        for e in range(2):
            pp, nz = replay_loader_rng(n, 100, D)
            perms.append(pp)
            noises.append(nz)
        opt = torch.optim.Adam(t.netG.parameters(), lr=1e-3, weight_decay=1e-6)
        losses, grads, ws = [], [], []
        Xt = torch.from_numpy(X)
        t.netG.train()
        for e in range(2):
            for b in range(0, n, 100):
                idx = torch.from_numpy(perms[e][b:b + 100].astype(np.int64))
                data = Xt[idx] + jitter * torch.from_numpy(noises[e][b:b + 100])
                opt.zero_grad()
                loss = -t.netG.log_probs(data).mean()
                loss.backward()
                grads.append(pack_grads(t.netG))
                opt.step()
                losses.append(loss.item())
                ws.append(pack_state_dict(t.netG))
        out.update(X=X, jitter=jitter, perms=np.stack(perms), noises=np.stack(noises), losses=np.array(losses),
                   grads=np.stack(grads[:2]), ws=np.stack([ws[0], ws[-1]]), lp1=t.log_probs(x, to_numpy=True))
        np.savez_compressed(os.path.join(OUT, 'base_gennormal_%s_d%d.npz' % (flow, D)), **out)
        print('G10 base', flow, D, 'loss0', losses[0], 'lp0', out['lp0'][:2])


# ----------------------------------------------------------------------------------------------
# G11: fast/slow hierarchy, FastSlowNVP (networks.py:86-150, :350-380; Trainer(num_slow=...), trainer.py:85-88): a slow NVP on
# the first num_slow dims, a fast NVP on the rest, then one coupling (hidden 64, 1 layer) conditioned on the slow block
# ----------------------------------------------------------------------------------------------
def gen_fastslow():
    for S, F in [(2, 3), (5, 5), (4, 12)]:
        D = S + F
        np.random.seed(0)
        torch.manual_seed(12)
        t = Trainer(D, num_slow=S, hidden_dim=16, num_blocks=3, num_layers=1, flow='nvp', log_dir=None, learning_rate=1e-3,
                    log_level=logging.WARNING)
        out = {'S': S, 'F': F, 'D': D, 'H': 16, 'B': 3, 'L': 1, 'keys': np.array(list(t.netG.state_dict().keys())),
               'shapes': np.array([str(tuple(v.shape)) for v in t.netG.state_dict().values()])}
        out['w0'] = pack_state_dict(t.netG)
        x = np.random.normal(size=(64, D)).astype(np.float32)
        n, jitter = 230, 0.02
        X = np.random.uniform(-1, 1, size=(n, D)).astype(np.float32)
        torch.manual_seed(11)
        perms, noises = [], []
        for e in range(2):
            pp, nz = replay_loader_rng(n, 100, D)
            perms.append(pp)
            noises.append(nz)
        for tag in ('init', 'trained'):
            if tag == 'trained':
                opt = torch.optim.Adam(t.netG.parameters(), lr=1e-3, weight_decay=1e-6)
                losses, grads, ws = [], [], []
                Xt = torch.from_numpy(X)
                t.netG.train()
                for e in range(2):
                    for b in range(0, n, 100):
                        idx = torch.from_numpy(perms[e][b:b + 100].astype(np.int64))
                        data = Xt[idx] + jitter * torch.from_numpy(noises[e][b:b + 100])
                        opt.zero_grad()
                        loss = -t.netG.log_probs(data).mean()
                        loss.backward()
                        grads.append(pack_grads(t.netG))
                        opt.step()
                        losses.append(loss.item())
                        ws.append(pack_state_dict(t.netG))
                out.update(X=X, jitter=jitter, perms=np.stack(perms), noises=np.stack(noises), losses=np.array(losses),
                           grads=np.stack(grads[:2]), ws=np.stack([ws[0], ws[-1]]))
            z, ldf = t.forward(x, to_numpy=True)
            xb, ldi = t.inverse(z, to_numpy=True)
            lp = t.log_probs(x, to_numpy=True)
            out.update({'x': x, 'w_' + tag: pack_state_dict(t.netG), 'z_' + tag: z, 'ldf_' + tag: ldf, 'xb_' + tag: xb,
                        'ldi_' + tag: ldi, 'lp_' + tag: lp})
        np.savez_compressed(os.path.join(OUT, 'fastslow_s%d_f%d.npz' % (S, F)), **out)
        print('G11 fastslow', S, F, 'nparams', out['w0'].size, 'loss0', out['losses'][0], list(out['keys'][-6:]))


# ----------------------------------------------------------------------------------------------
# G12: FastSlowSpline (networks.py:718-731): spline stages (fast: hidden 16 always) + the hidden-64 NVP coupling
# ----------------------------------------------------------------------------------------------
def gen_fastslow_spline():
    for S, F in [(2, 3), (5, 4)]:
        D = S + F
        np.random.seed(0)
        torch.manual_seed(13)
        t = Trainer(D, num_slow=S, hidden_dim=16, num_blocks=3, flow='spline', log_dir=None, learning_rate=1e-3,
                    log_level=logging.WARNING)
        out = {'S': S, 'F': F, 'D': D, 'H': 16, 'B': 3, 'keys': np.array(list(t.netG.state_dict().keys())),
               'P_fast': np.stack([f.P.numpy() for f in t.netG.fast_flow.flows if hasattr(f, 'P')]).astype(np.float32),
               'P_slow': np.stack([f.P.numpy() for f in t.netG.slow_flow.flows if hasattr(f, 'P')]).astype(np.float32)}
        out['w_raw'] = pack_state_dict(t.netG)
        x_first = np.random.uniform(-1, 1, size=(100, D)).astype(np.float32)
        zf_, ldf_ = t.forward(x_first, to_numpy=True)     # ActNorm data-dependent init of both stages
        out.update(x_first=x_first, z_first=zf_, ld_first=ldf_, w_init=pack_state_dict(t.netG))
        x = np.random.uniform(-1, 1, size=(64, D)).astype(np.float32)
        n, jitter = 230, 0.02
        X = np.random.uniform(-1, 1, size=(n, D)).astype(np.float32)
        torch.manual_seed(11)
        perms, noises = [], []
        for e in range(2):
            pp, nz = replay_loader_rng(n, 100, D)
            perms.append(pp)
            noises.append(nz)
        for tag in ('init', 'trained'):
            if tag == 'trained':
                opt = torch.optim.Adam(t.netG.parameters(), lr=1e-3, weight_decay=1e-6)
                losses, grads, ws = [], [], []
                Xt = torch.from_numpy(X)
                for e in range(2):
                    for b in range(0, n, 100):
                        idx = torch.from_numpy(perms[e][b:b + 100].astype(np.int64))
                        data = Xt[idx] + jitter * torch.from_numpy(noises[e][b:b + 100])
                        opt.zero_grad()
                        loss = -t.netG.log_probs(data).mean()
                        loss.backward()
                        grads.append(pack_grads(t.netG))
                        opt.step()
                        losses.append(loss.item())
                        ws.append(pack_state_dict(t.netG))
                out.update(X=X, jitter=jitter, perms=np.stack(perms), noises=np.stack(noises), losses=np.array(losses),
                           grads=np.stack(grads[:2]), ws=np.stack([ws[0], ws[-1]]))
            z, ldz = t.forward(x, to_numpy=True)
            xb, ldi = t.inverse(z, to_numpy=True)
            out.update({'x': x, 'w_' + tag: pack_state_dict(t.netG), 'z_' + tag: z, 'ldf_' + tag: ldz, 'xb_' + tag: xb,
                        'ldi_' + tag: ldi, 'lp_' + tag: t.log_probs(x, to_numpy=True)})
        np.savez_compressed(os.path.join(OUT, 'fastslowspline_s%d_f%d.npz' % (S, F)), **out)
        print('G12 fastslow spline', S, F, 'nparams', out['w_raw'].size, 'loss0', out['losses'][0])


# ----------------------------------------------------------------------------------------------
# G13: 'choleksy' flow (networks.py:162-239)
# ----------------------------------------------------------------------------------------------
def gen_cholesky():
    for D in (2, 5, 20):
        np.random.seed(0)
        torch.manual_seed(14)
        t = Trainer(D, flow='choleksy', log_dir=None, learning_rate=1e-3, log_level=logging.WARNING)
        with torch.no_grad():   # move off the identity so that every parameter matters
            for p_ in t.netG.parameters():
                p_.add_(0.3 * torch.randn_like(p_))
        out = {'D': D, 'keys': np.array(list(t.netG.state_dict().keys())), 'w0': pack_state_dict(t.netG)}
        x = np.random.normal(size=(64, D)).astype(np.float32)
        z, ldf = t.forward(x, to_numpy=True)
        xb, ldi = t.inverse(z, to_numpy=True)
        out.update(x=x, z=z, ldf=ldf, xb=xb, ldi=ldi, lp=t.log_probs(x, to_numpy=True))
        n, jitter = 230, 0.02
        X = np.random.uniform(-1, 1, size=(n, D)).astype(np.float32)
        torch.manual_seed(11)
        perms, noises = [], []
        for e in range(2):
            pp, nz = replay_loader_rng(n, 100, D)
            perms.append(pp)
            noises.append(nz)
        opt = torch.optim.Adam(t.netG.parameters(), lr=1e-3, weight_decay=1e-6)
        losses, grads, ws = [], [], []
        Xt = torch.from_numpy(X)
        for e in range(2):
            for b in range(0, n, 100):
                idx = torch.from_numpy(perms[e][b:b + 100].astype(np.int64))
                data = Xt[idx] + jitter * torch.from_numpy(noises[e][b:b + 100])
                opt.zero_grad()
                loss = -t.netG.log_probs(data).mean()
                loss.backward()
                grads.append(pack_grads(t.netG))
                opt.step()
                losses.append(loss.item())
                ws.append(pack_state_dict(t.netG))
        out.update(X=X, jitter=jitter, perms=np.stack(perms), noises=np.stack(noises), losses=np.array(losses), grads=np.stack(grads[:2]),
                   ws=np.stack([ws[0], ws[-1]]))
        np.savez_compressed(os.path.join(OUT, 'cholesky_d%d.npz' % D), **out)
        print('G13 cholesky', D, list(out['keys']), 'loss0', losses[0])


if __name__ == '__main__':
    if 'cholesky' in (sys.argv[1:] or ['cholesky']):
        gen_cholesky()
    which = sys.argv[1:] or ['flow', 'like', 'like2', 'train', 'trainrun', 'mcmc', 'mcmcspline', 'nested', 'scale', 'spline', 'base',
                             'fastslow', 'fastslowspline']
    if 'fastslowspline' in which:
        gen_fastslow_spline()
    if 'fastslow' in which:
        gen_fastslow()
    if 'base' in which:
        gen_base_dist()
    if 'spline' in which:
        gen_spline()
    if 'scale' in which:
        gen_scale_variants()
    if 'flow' in which:
        gen_flow()
    if 'like' in which:
        gen_like()
    if 'like2' in which:
        gen_like2()
    if 'train' in which:
        gen_train_steps()
    if 'trainrun' in which:
        gen_train_run()
    if 'mcmc' in which:
        gen_mcmc()
    if 'mcmcspline' in which:
        gen_mcmc_spline()
    if 'nested' in which:
        gen_nested()
    if 'formats' in which:
        gen_formats()
