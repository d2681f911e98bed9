"""Sampler base: the reference's nnest.sampler.Sampler contract (nnest/sampler.py:29-222) with the batched
latent-space Metropolis proposal `_mcmc_sample` (nnest/sampler.py:229-463) running as ONE persistent HIP
kernel launch per batch when the likelihood is one of the analytic ones the kernels know
(nnest_mh_constrained_steps, include/nnest_hip.h), and as GPU flow passes + host likelihood callbacks
otherwise (any user `loglike`, the reference's plugin protocol).

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL).  State is replicated, the chains of
a batch are sharded over ranks, and the only data-path collective is one all-gather of the chain endpoints
per batch (the reference's mpi4py gather+bcast of full histories, nnest/nested.py:416-427, shrunk to what is
consumed).
"""
import itertools
import json
import logging
import os

import numpy as np
import torch

from . import _lib
from .utils import create_logger, get_or_create_run_dir, write_rows_e5


def detect_linear_scale(transform, x_dim):
    """transform(x) == s * x for a scalar s?  (examples/nested/run.py:25-42 only uses such transforms.)
    Returns s or None.  Probing is on a few random rows; the fused path is additionally verified against the
    host callable before it is enabled (Sampler._verify_fused)."""
    if transform is None:
        return 1.0
    x = np.random.RandomState(12345).uniform(-1, 1, size=(4, x_dim))
    try:
        y = np.asarray(transform(x.copy()), dtype=np.float64)
    except Exception:
        return None
    if y.shape != x.shape:
        return None
    s = float(np.asarray(transform(np.ones((1, x_dim))), dtype=np.float64)[0, 0])  # exact, no division
    return s if np.allclose(y, s * x, rtol=1e-12, atol=0) else None


class Sampler(object):

    def __init__(self,
                 x_dim,
                 loglike,
                 transform=None,
                 prior=None,
                 append_run_num=True,
                 hidden_dim=16,
                 num_slow=0,
                 num_derived=0,
                 batch_size=100,
                 flow='spline',
                 num_blocks=3,
                 num_layers=1,
                 learning_rate=0.001,
                 log_dir='logs/test',
                 resume=True,
                 use_gpu=True,
                 base_dist=None,
                 scale='',
                 trainer=None,
                 transform_prior=True,
                 oversample_rate=-1,
                 log_level=logging.INFO,
                 param_names=None,
                 fused=True,
                 mcmc_history=False,
                 mcmc_proposal='mh'):
        # mcmc_proposal (not in the reference, whose _mcmc_sample proposes random-walk Metropolis moves only, sampler.py:310-316):
        # 'mh' = that step; 'slice' = the build-defined slice proposal in latent space (BASELINE north_star; nnest_slice_steps,
        # include/nnest_hip.h; parity unpinned) on the fused kernel path
        if mcmc_proposal not in ('mh', 'slice'):
            raise ValueError("mcmc_proposal=%r: 'mh' (the reference's step) or 'slice' (build-defined)" % (mcmc_proposal,))
        self.mcmc_proposal = mcmc_proposal
        self.x_dim = x_dim
        self.num_derived = num_derived
        self.num_params = x_dim + num_derived
        assert x_dim > num_slow                       # sampler.py:87
        self.num_slow = num_slow
        self.num_fast = x_dim - num_slow
        self.param_names = param_names
        if self.param_names is not None:
            assert len(param_names) == self.num_params
        self.oversample_rate = oversample_rate if oversample_rate > 0 else self.num_fast / self.x_dim   # sampler.py:95
        self.mcmc_history = mcmc_history
        self._user_loglike = loglike
        self._user_prior = prior
        self._transform_prior = transform_prior

        self._user_transform = transform
        self.transform = self._checked_transform if transform is not None else (lambda x: x)
        self._linear_scale = detect_linear_scale(transform, x_dim)
        self.loglike = self._checked_loglike
        sample_prior = getattr(prior, 'sample', None)
        self.sample_prior = sample_prior if callable(sample_prior) else None
        self.prior = self._checked_prior

        # process group (the reference: mpi4py, sampler.py:165-177)
        self.use_mpi = False
        self.mpi_size, self.mpi_rank = 1, 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.mpi_size = torch.distributed.get_world_size()
            self.mpi_rank = torch.distributed.get_rank()
            # an initialised process group selects the distributed code path, also with ONE rank (the degenerate case runs
            # the same collectives; it is how the RCCL branch is exercised on a one-GPU box)
            self.use_mpi = True
        self.single_or_primary_process = (not self.use_mpi) or self.mpi_rank == 0

        # only the primary process owns a run directory: the others neither write it nor read checkpoints from it (the
        # reference lets every MPI rank read the checkpoint files, sampler.py:179-185; here rank 0 broadcasts the resumed state)
        if self.single_or_primary_process:
            self.logs = get_or_create_run_dir(log_dir, append_run_num=append_run_num)
            self.log_dir = self.logs['run_dir']
        else:
            self.logs = None
            self.log_dir = None
        if self.single_or_primary_process:
            args = {k: str(v) for k, v in dict(x_dim=x_dim, num_derived=num_derived, hidden_dim=hidden_dim, flow=flow,
                                               num_blocks=num_blocks, num_layers=num_layers, batch_size=batch_size,
                                               learning_rate=learning_rate, log_dir=self.log_dir, scale=scale,
                                               transform_prior=transform_prior, param_names=param_names).items()}
            with open(os.path.join(self.logs['info'], 'params.txt'), 'w') as f:
                json.dump(args, f, indent=4)

        self.resume = resume
        self.logger = create_logger(__name__, level=log_level)

        if trainer is None:
            from .trainer import Trainer
            self.trainer = Trainer(x_dim, hidden_dim=hidden_dim, num_slow=num_slow, batch_size=batch_size, flow=flow,
                                   num_blocks=num_blocks, num_layers=num_layers, learning_rate=learning_rate,
                                   log_dir=self.log_dir if self.single_or_primary_process else None,
                                   log=self.single_or_primary_process, use_gpu=use_gpu,
                                   base_dist=base_dist, scale=scale, log_level=log_level)
        else:
            self.trainer = trainer

        if self.single_or_primary_process:
            self.logger.info('Num base params [%d]' % self.x_dim)
            self.logger.info('Num derived params [%d]' % self.num_derived)
            self.logger.info('Total params [%d]' % self.num_params)

        self.total_accepted = 0
        self.total_rejected = 0
        self.total_calls = 0
        self.total_fast_calls = 0

        # can the whole proposal loop run inside the HIP kernel?
        self._fused_like_id = None
        self._fused_like_params = ()
        if fused:
            self._fused_like_id = self._fused_eligibility()

    # ---- the user's callables behind shape checks (sampler.py:97-163) ----------------------------------------------
    def _rows(self, x):
        """[D] or [N, D] (array or list) -> [N, D]"""
        x = np.asarray(x) if isinstance(x, list) else x
        if x.ndim == 1:
            assert x.shape[0] == self.x_dim
            x = x[None, :]
        return x

    def _checked_transform(self, x):
        return self._user_transform(self._rows(x))

    def _checked_loglike(self, x):
        """loglike(transform(x)) -> (logl float64 [N], derived [N, num_derived]); counts the rows as likelihood calls;
        non-finite values become -1e100 (sampler.py:110-133)"""
        x = self._rows(x)
        n = x.shape[0]
        out = self._user_loglike(self.transform(x))
        self.total_calls += n
        logl, derived = out if isinstance(out, tuple) else (out, None)
        logl = np.array(logl, dtype=np.float64, ndmin=1)   # a copy, float64: -1e100 is not representable in float32
        logl[~np.isfinite(logl)] = -1e100
        if derived is None:
            derived = np.empty((n, 0))
        elif np.ndim(derived) == 1:
            raise ValueError('Derived should have dimensions (batch size, num derived params)')
        derived = np.asarray(derived)
        if derived.shape[1] != self.num_derived:
            raise ValueError('Is the number of derived parameters correct?')
        return logl, derived

    def _checked_prior(self, x):
        """log prior density of every row (sampler.py:137-163): the prior protocol is one row -> one number"""
        x = self._rows(x)
        prior = self._user_prior
        if prior is None:
            return np.zeros(x.shape[0])
        rows = getattr(prior, 'log_prob_rows', None)   # whole-batch form of the same rule, when the prior has one
        if callable(rows):
            return rows(self.transform(x) if self._transform_prior else x)
        if self._transform_prior:   # one row at a time through the transform, as the reference calls it (sampler.py:159)
            return np.array([prior(self.transform(r)) for r in x])
        return np.array([prior(r) for r in x])

    # ---- fused-path eligibility -----------------------------------------------------------------------
    def _fused_eligibility(self):
        like = self._user_loglike
        like_id = getattr(like, 'hip_like_id', None)
        netG = getattr(self.trainer, 'netG', None)
        if like_id is None or getattr(netG, 'mh_steps', None) is None or self._linear_scale is None or self.num_derived != 0:
            return None
        prior = self._user_prior
        if prior is None or self._transform_prior or not getattr(prior, 'is_unit_box', lambda: False)():
            return None
        # verify the kernel's likelihood against the host callable on a few points before trusting it
        from . import flow
        x = np.random.RandomState(4321).uniform(-1, 1, size=(32, self.x_dim)).astype(np.float32)
        self._fused_like_params = tuple(getattr(like, 'hip_like_params', ()) or ())
        dev = flow.loglike(like_id, x, self._linear_scale, device=netG.device, like_params=self._fused_like_params).cpu().numpy()
        host = np.asarray(like(self._linear_scale * x.astype(np.float64)), dtype=np.float64)
        if not np.allclose(dev, host, rtol=1e-5, atol=1e-4):
            self.logger.warning('fused likelihood id %d disagrees with the host callable; using the host protocol' % like_id)
            return None
        return like_id

    def _next_seed(self):
        """64-bit seed for the in-kernel Philox streams.  Drawn from a PRIVATE generator seeded once from the seed of torch's
        CPU generator (torch.manual_seed): the global generator is never advanced by the build's own bookkeeping, so the
        host protocol consumes torch / numpy randomness draw for draw as the reference's loop does (nested.py:398-456;
        tests/test_reference_trajectory.py holds the two drivers to the same trajectory)."""
        gen = getattr(self, '_seed_gen', None)
        if gen is None:
            gen = self._seed_gen = torch.Generator(device='cpu')
            gen.manual_seed(torch.initial_seed())
        return int(torch.empty((), dtype=torch.int64).random_(generator=gen).item())

    # ---- the batched proposal ---------------------------------------------------------------------------
    def _mcmc_sample(self,
                     mcmc_steps,
                     step_size=0.0,
                     dynamic_step_size=False,
                     num_chains=1,
                     init_samples=None,
                     init_loglikes=None,
                     init_derived=None,
                     loglstar=None,
                     show_progress=False,
                     max_start_tries=100,
                     output_interval=None,
                     stats_interval=None,
                     plot_trace=False,
                     prior_volume_steps=1,
                     walker_offset=0,
                     seed=None,
                     form=None):
        """Returns (samples, latent_samples, derived_samples, loglikes, scale, ncall) shaped as the reference
        (chain, step, dim), sampler.py:455-463.  On the fused path the step axis holds only the first and the
        last state unless the sampler was built with mcmc_history=True (nested.py:432-437 reads only those)."""
        if step_size <= 0.0:
            step_size = 2 / self.x_dim ** 0.5
        fused = (self._fused_like_id is not None and init_samples is not None and init_loglikes is not None
                 and prior_volume_steps == 1)   # loglstar None = the unconstrained branch, also in the kernel
        if fused:
            return self._mcmc_sample_fused(mcmc_steps, step_size, dynamic_step_size, init_samples, init_loglikes,
                                           loglstar, walker_offset, seed, form)
        return self._mcmc_sample_host(mcmc_steps, step_size, dynamic_step_size, num_chains, init_samples,
                                      init_loglikes, init_derived, loglstar, max_start_tries, prior_volume_steps)

    def _fused_launch(self, mcmc_steps, step_size, dynamic, init_samples, init_loglikes, loglstar, walker_offset, seed,
                      form=None):
        """One K4 launch from live points: returns (res, z0, z, logl) device-side (no host copies)."""
        netG = self.trainer.netG
        z, _ = netG.forward(init_samples)                               # sampler.py:264
        logl = torch.as_tensor(np.ascontiguousarray(init_loglikes, dtype=np.float64)).to(netG.device)
        z0 = z.clone()
        kw = dict(seed=self._next_seed() if seed is None else seed, walker_offset=walker_offset, history=self.mcmc_history,
                  like_params=self._fused_like_params, form=form)
        star = None if loglstar is None else float(loglstar)
        args = (self._fused_like_id, self._linear_scale, z, logl, star, float(step_size), int(mcmc_steps))
        # dynamic: the reference's rule over the whole batch (sampler.py:422-431), `mcmc_step_lag` steps behind (0 = exactly
        # the reference; the default keeps the grid-wide wait off the step); populations too large for a
        # resident grid fall back to the per-16-walker rule
        mode = 'batch' if dynamic and getattr(self, '_batch_rule_ok', True) else ('group' if dynamic else False)
        lag = getattr(self, 'mcmc_step_lag', None)
        if lag is None and int(mcmc_steps) < 100:
            lag = 0   # short chains: the lag would be a sizeable part of the launch, and the exact rule costs microseconds here
        if getattr(self, 'mcmc_step_warm', None) is not None:
            kw['warm'] = int(self.mcmc_step_warm)
        try:
            res = netG.mh_steps(*args, dynamic=mode, lag=lag, **kw)
        except _lib.NnestHipError as e:
            if e.code != _lib.NNEST_E_UNSUPPORTED or (mode != 'batch' and form is None):
                raise
            # refused: the batch-wide rule on a grid that may not be resident, or a pinned form that does not apply to this
            # shape / rule, or exact warm-up steps asked of a form that has none (they belong to the solo form).
            res = None
            if mode == 'batch' and kw.get('warm', 0) > 0:
                # first the same rule WITHOUT the warm-up (the batch rule itself may well be available)
                try:
                    res = netG.mh_steps(*args, dynamic=mode, lag=lag, **dict(kw, warm=0))
                    if not getattr(self, '_warned_warm', False):
                        self.logger.warning('mcmc_step_warm=%d dropped: the kernel form for %d walkers has no exact warm-up steps' % (kw['warm'], z.shape[0]))
                        self._warned_warm = True
                    self.mcmc_step_warm = 0
                except _lib.NnestHipError as e2:
                    if e2.code != _lib.NNEST_E_UNSUPPORTED:
                        raise
            if res is not None:
                return res, z0, z, logl
            # Next: the per-group rule, with the form the library picks for it.
            if mode == 'batch':
                self.logger.warning('batch-wide step rule not available for %d walkers (%s); using the per-group rule' % (z.shape[0], e))
                self._batch_rule_ok = False
                mode = 'group'
            kw['form'] = None
            res = netG.mh_steps(*args, dynamic=mode, lag=lag, **kw)
        return res, z0, z, logl

    def _mcmc_sample_fused(self, mcmc_steps, step_size, dynamic, init_samples, init_loglikes, loglstar, walker_offset,
                           seed, form=None):
        netG = self.trainer.netG
        C = init_samples.shape[0]
        res, z0, z, logl = self._fused_launch(mcmc_steps, step_size, dynamic, init_samples, init_loglikes, loglstar,
                                              walker_offset, seed, form)
        ncall = int(res['n_call'].sum().item())
        nacc = int(res['n_accept'].sum().item())
        netG.check_sync(res)
        self.total_calls += ncall
        self.total_accepted += nacc
        self.total_rejected += C * int(mcmc_steps) - nacc
        if self.mcmc_history:
            samples = res['hist_x'].cpu().numpy()
            loglikes = res['hist_logl'].cpu().numpy()
            latent = np.stack([z0.cpu().numpy(), z.cpu().numpy()], axis=1)
        else:
            x0, _ = netG.inverse(z0)                                    # sampler.py:266
            samples = torch.stack([x0, res['x']], dim=1).cpu().numpy()
            loglikes = np.stack([np.asarray(init_loglikes, dtype=np.float64), logl.cpu().numpy()], axis=1)
            latent = torch.stack([z0, z], dim=1).cpu().numpy()
        derived = np.empty((C, samples.shape[1], 0))
        scale = float(res['scale'].mean().item()) if dynamic else float(step_size)
        return samples, latent, derived, loglikes, scale, ncall

    def _mcmc_endpoints_fused(self, mcmc_steps, step_size, dynamic, init_samples, init_loglikes, loglstar, walker_offset, seed,
                              form=None):
        """What the nested-sampling loop consumes of a batch (nested.py:432-437) -- per chain its end x, its end logL and whether
        every coordinate moved (the reference's test of a usable chain, evaluated here, where the start x exists) -- as ONE
        device tensor [C, D + 2] float64, so that the per-batch all-gather (C2) runs on device memory and carries no start
        points (round 2 gathered [C, 2 D + 1])."""
        netG = self.trainer.netG
        C = init_samples.shape[0]
        if self.mcmc_proposal == 'slice':
            # BUILD-DEFINED (the reference has no slice proposal): `mcmc_steps` slice-sampling updates per chain along random
            # directions of latent space.  Initial bracket: twice the Metropolis step (z + t * 2 step_size * eps, |eps| ~ sqrt(D): with
            # the default step_size = 1 / sqrt(D) a bracket of length ~2, the chord of a latent ball of radius sqrt(D) through a point
            # near its surface in a random direction); every update moves (no rejection), so the scale is not adapted -- stepping
            # out and shrinkage find the slice's extent
            if loglstar is None:
                raise NotImplementedError("mcmc_proposal='slice' samples under the hard constraint logL > loglstar only")
            z, _ = netG.forward(init_samples)
            logl = torch.as_tensor(np.asarray(init_loglikes, dtype=np.float64), device=z.device).contiguous()
            res = netG.slice_steps(self._fused_like_id, self._linear_scale, z, logl, float(loglstar), 2.0 * float(step_size),
                                   int(mcmc_steps), seed=self._next_seed() if seed is None else seed, walker_offset=walker_offset,
                                   like_params=self._fused_like_params)
            ends = torch.cat([res['x'].double(), logl[:, None], res['moved'][:, None].double()], dim=1)
            counts = torch.stack([res['n_call'].sum(), res['n_move'].sum()]).cpu()
            ncall, nmove = int(counts[0]), int(counts[1])
            self.total_calls += ncall
            self.total_accepted += nmove
            self.total_rejected += C * int(mcmc_steps) - nmove
            return ends, float(step_size), ncall
        res, z0, z, logl = self._fused_launch(mcmc_steps, step_size, dynamic, init_samples, init_loglikes, loglstar,
                                              walker_offset, seed, form)
        # "every coordinate moved" (nested.py:432: samples[:, 0] != samples[:, -1] in every dimension): the kernel's own test of the
        # chain's last x against its first x = f^-1(z_0), coordinate by coordinate (NNEST_MH_ALL_MOVED; round 4 took "accepted at
        # least once" for it, which differs once a proposal is so small that a coordinate of x does not change in float32)
        moved = res['moved']
        ends = torch.cat([res['x'].double(), logl[:, None], moved[:, None].double()], dim=1)
        counts = torch.stack([res['n_call'].sum(), res['n_accept'].sum()]).cpu()   # one small copy; orders the stream too
        netG.check_sync(res)
        ncall, nacc = int(counts[0]), int(counts[1])
        self.total_calls += ncall
        self.total_accepted += nacc
        self.total_rejected += C * int(mcmc_steps) - nacc
        scale = float(res['scale'].mean().item()) if dynamic else float(step_size)
        return ends, scale, ncall

    class _StepScale(object):
        """the proposal scale under the reference's adaptation rule (sampler.py:422-431): a step in which more than half of
        the chains moved votes up, any other step votes down; the scale follows whichever side leads"""

        def __init__(self, value):
            self.value, self.up, self.down = value, 0, 0

        def vote(self, moved, chains):
            if 2 * moved > chains:
                self.up += 1
            else:
                self.down += 1
            if self.up > self.down:
                self.value *= np.exp(1. / (1 + self.up))
            elif self.up < self.down:
                self.value /= np.exp(1. / (1 + self.down))

    def _propose(self, z, scale):
        """z + scale * N(0, I); with a fast/slow hierarchy a fraction `oversample_rate` of the proposals leaves the slow
        block alone (sampler.py:310-316, :377-382).  Returns (z', fast?)"""
        dz = torch.randn_like(z) * scale
        fast = self.num_slow > 0 and np.random.uniform() < self.oversample_rate
        if fast:
            dz[:, 0:self.num_slow] = 0.0
        return z + dz, fast

    @staticmethod
    def _metropolis_mask(log_ratio):
        """u < min(1, e^log_ratio) row by row (sampler.py:334-336): bool numpy array"""
        log_ratio = log_ratio if torch.is_tensor(log_ratio) else torch.as_tensor(log_ratio)
        return (torch.rand(log_ratio.shape) < log_ratio.exp().clamp(max=1)).numpy().astype(bool)

    def _mcmc_sample_host(self, mcmc_steps, step_size, dynamic, num_chains, init_samples, init_loglikes, init_derived,
                          loglstar, max_start_tries, prior_volume_steps):
        """The reference's step loop (sampler.py:246-463) with the flow passes on the GPU and the user's
        likelihood / prior callables on the host: the path for likelihoods the kernels do not know."""
        tr = self.trainer
        tr.netG.eval()
        ncall = 0
        if init_samples is not None:
            num_chains = init_samples.shape[0]
            z, _ = tr.forward(init_samples)
            x = tr.get_samples(z, to_numpy=True)
            if init_loglikes is None or init_derived is None:
                logl, derived = self.loglike(x)
                ncall += num_chains
            else:
                logl, derived = np.array(init_loglikes, dtype=np.float64), init_derived
            logl_prior = self.prior(x)
        else:   # sampler.py:270-284: draw from the flow's base until every chain starts at finite density
            for attempt in range(max_start_tries):
                z = tr.get_prior_samples(num_chains)
                x = tr.get_samples(z, to_numpy=True)
                logl, derived = self.loglike(x)
                ncall += num_chains
                logl_prior = self.prior(x)
                if np.all(logl > -1e30) and np.all(logl_prior > -1e30):
                    break
            else:
                raise Exception('Could not find starting value')
        z = z if torch.is_tensor(z) else torch.as_tensor(z)
        hist = dict(x=[x], z=[z.cpu().numpy()], derived=[derived], logl=[logl])
        step = self._StepScale(step_size)
        for it in range(mcmc_steps):
            _, log_det_J = tr.inverse(z)                 # sampler.py:295 (x itself is carried: value-identical)
            fast = False
            if loglstar is not None:
                # hard constraint logl > loglstar.  Up to prior_volume_steps proposals from the CURRENT z; a chain keeps the
                # last one that passed the prior box and the Jacobian test (sampler.py:300-345), then its likelihood decides.
                z_prime, x_prime = z, x
                moved = np.zeros(num_chains, dtype=bool)
                for _ in range(prior_volume_steps):
                    z_try, fast = self._propose(z, step.value)
                    try:
                        x_try_t, log_det_J_try = tr.inverse(z_try)
                    except ValueError:           # a flow may refuse an inverse (sampler.py:322-324): proposal skipped
                        continue
                    x_try = x_try_t.cpu().numpy()
                    log_ratio = (log_det_J_try - log_det_J).cpu()
                    log_ratio[torch.as_tensor(self.prior(x_try) < -1e30)] = -np.inf
                    ok = self._metropolis_mask(log_ratio)
                    z_prime = torch.where(torch.as_tensor(ok, device=z.device)[:, None], z_try, z_prime)
                    x_prime = np.where(ok[:, None], x_try, x_prime)
                    moved |= ok
                mask = moved
                logl_prior_prime = self.prior(x_prime)
                logl_prime = np.array(logl, dtype=np.float64, copy=True)
                derived_prime = np.copy(derived)
                cand = np.flatnonzero(mask)
                if cand.size:                    # the likelihood is evaluated only where the cheap tests passed (:358-368)
                    lp, der = self.loglike(x_prime[cand])
                    ncall += cand.size
                    if fast:
                        self.total_fast_calls += cand.size
                    good = np.isfinite(lp) & (lp > loglstar)
                    logl_prime[cand[good]] = lp[good]
                    derived_prime[cand[good]] = der[good]
                    mask[cand[~good]] = False
            else:
                # no constraint: likelihood and prior enter the Metropolis ratio (sampler.py:372-413)
                z_prime, fast = self._propose(z, step.value)
                try:
                    x_prime_t, log_det_J_prime = tr.inverse(z_prime)
                except ValueError:
                    continue
                x_prime = x_prime_t.cpu().numpy()
                ncall += num_chains
                if fast:
                    self.total_fast_calls += num_chains
                logl_prime, derived_prime = self.loglike(x_prime)
                logl_prior_prime = self.prior(x_prime)
                mask = self._metropolis_mask((log_det_J_prime - log_det_J).cpu() + torch.as_tensor(logl_prime - logl)
                                             + torch.as_tensor(logl_prior_prime - logl_prior))
            n_moved = int(mask.sum())
            self.total_accepted += n_moved
            self.total_rejected += num_chains - n_moved
            if dynamic:
                step.vote(n_moved, num_chains)
            sel = mask[:, None]
            logl = np.where(mask, logl_prime, logl)
            logl_prior = np.where(mask, logl_prior_prime, logl_prior)
            z = torch.where(torch.as_tensor(sel, device=z.device), z_prime, z)
            x = np.where(sel, x_prime, x)
            derived = np.where(sel, derived_prime, derived)
            for k, v in (('x', x), ('z', z.cpu().numpy()), ('derived', derived), ('logl', logl)):
                hist[k].append(v)
        chain_major = lambda a: np.moveaxis(np.array(a), 0, 1)   # (step, chain, ...) -> (chain, step, ...)  sampler.py:455-459
        return (chain_major(hist['x']), chain_major(hist['z']), chain_major(hist['derived']), chain_major(hist['logl']),
                step.value, ncall)

    # ---- prior rejection (sampler.py:529-543) ---------------------------------------------------------------
    def _rejection_prior_sample(self, loglstar, num_trials=None):
        if num_trials is None and self._fused_like_id is not None:
            # The reference draws one prior sample per likelihood call until one passes (sampler.py:531-538).  The
            # draws are independent, so the same rule is applied to a block per launch of the likelihood kernel: the
            # first candidate above loglstar is returned and ncall counts the candidates examined up to and including it.
            # Round 4: candidates a launch evaluated but the caller did not reach (they lie behind the accepted one) are independent
            # prior draws like any other, so they are KEPT and examined first by the next call (against its higher loglstar), each
            # exactly once and in order: one launch + one read-back now serves several iterations of the outer loop instead of one
            # (5 264 calls of 0.11 ms in a config-2 run, latency-bound).
            # Each block keeps the (sorted) indices of its candidates above the threshold AT ITS CREATION with their float64 likelihoods
            # from one vectorised host call: the threshold only rises, so every later hit is among them, and a call walks a pointer
            # instead of searching the block (config 3: 8 371 calls, 0.59 s of a 1.9 s run before).
            from . import flow
            ncall = 0
            cache = getattr(self, '_prior_cache', None)
            while True:
                if cache is None or cache['pos'] >= len(cache['logl']) or loglstar < cache['star0']:   # (a LOWER threshold than the block was indexed for: start afresh)
                    block = getattr(self, '_prior_block', 256)
                    x = self.sample_prior(block)
                    logl = flow.loglike(self._fused_like_id, x, self._linear_scale, device=self.trainer.netG.device,
                                        like_params=self._fused_like_params).cpu().numpy()
                    cand = np.flatnonzero(logl > loglstar)
                    l64 = d64 = None
                    if len(cand):   # the kernel works on float32(x); the stored value is the reference's float64 one
                        calls = self.total_calls
                        l64, d64 = self.loglike(x[cand])
                        self.total_calls = calls
                    cache = self._prior_cache = dict(x=x, logl=logl, pos=0, hits=0, cand=cand, k=0, l64=l64, d64=d64, star0=loglstar)
                x, logl, pos, cand, k = cache['x'], cache['logl'], cache['pos'], cache['cand'], cache['k']
                found = None
                while k < len(cand):
                    j = int(cand[k])
                    if j >= pos and logl[j] > loglstar and cache['l64'][k] > loglstar:
                        found = (j, k)
                        break
                    k += 1
                if found is not None:
                    j, k = found
                    self.total_calls += j + 1 - pos
                    cache['pos'], cache['k'] = j + 1, k + 1
                    cache['hits'] += 1
                    # the next block: ~ 16 acceptances' worth of candidates at the rate seen, bounded
                    self._prior_block = int(min(65536, max(256, 16 * (j + 1) / cache['hits'])))
                    return x[j:j + 1], cache['l64'][k:k + 1], cache['d64'][k:k + 1], ncall + j + 1 - pos
                self.total_calls += len(logl) - pos
                ncall += len(logl) - pos
                if cache['hits'] == 0:
                    self._prior_block = int(min(65536, 4 * len(logl)))
                cache = self._prior_cache = None
        if num_trials is not None:   # a fixed block of prior draws; ncall = expected draws per success (sampler.py:540-542)
            x = self.sample_prior(num_trials)
            logl, derived = self.loglike(x)
            return x, logl, derived, num_trials / np.sum(logl > loglstar)
        for ncall in itertools.count(1):   # one prior draw per likelihood call until one lies above loglstar (:531-538)
            x = self.sample_prior(1)
            logl, derived = self.loglike(x)
            if logl > loglstar:
                return x, logl, derived, ncall

    # ---- flow rejection (sampler.py:545-605) and density sampling (sampler.py:607-628) ----------------------------
    # The reference draws one candidate, inverts it, tests it, and loops.  Candidates are independent, so the same
    # chain of decisions is taken here over a block of candidates per launch: candidates are examined in order, the
    # first that passes every test is returned, and `ncall` counts -- as the reference does -- the candidates that
    # reached the likelihood test (box prior and Jacobian envelope passed) up to and including the accepted one.
    def _candidate_block(self, z):
        """x, log|dx/dz|, logl for a block of latent candidates; logl is None when the likelihood is a host callable."""
        netG = self.trainer.netG
        if self._fused_like_id is not None:
            x, ld, logl, inbox = netG.inverse_loglike(self._fused_like_id, self._linear_scale, z,
                                                      like_params=self._fused_like_params)
            return x.cpu().numpy().astype(np.float64), ld.cpu().numpy().astype(np.float64), logl.cpu().numpy(), \
                inbox.cpu().numpy().astype(bool)
        x, ld = netG.inverse(z)
        x = x.cpu().numpy().astype(np.float64)
        return x, ld.cpu().numpy().astype(np.float64), None, self.prior(x) > -1e30

    def _first_accepted(self, x, passed, logl, loglstar, strict):
        """Walk the candidates that passed the cheap tests in order; returns (index or None, likelihood calls made)."""
        idx = np.where(passed)[0]
        if logl is not None:                       # likelihood already evaluated in the launch
            good = logl[idx] > loglstar if strict else ~(np.isfinite(logl[idx]) & (logl[idx] < loglstar))
            hit = np.where(good)[0]
            if len(hit) == 0:
                self.total_calls += len(idx)
                return None, len(idx), None, None
            self.total_calls += int(hit[0]) + 1
            j = idx[hit[0]]
            return j, int(hit[0]) + 1, logl[j:j + 1], np.empty((1, 0))
        calls = 0
        for lo in range(0, len(idx), 16):          # host callable: evaluate in small groups, stop at the first hit
            grp = idx[lo:lo + 16]
            before = self.total_calls
            lg, dg = self.loglike(x[grp])
            good = lg > loglstar if strict else ~(np.isfinite(lg) & (lg < loglstar))
            hit = np.where(good)[0]
            if len(hit) > 0:
                # the reference evaluates one candidate per call and stops at the accepted one: the rest of the group is
                # not counted, neither in the returned ncall nor in total_calls
                self.total_calls = before + int(hit[0]) + 1
                return grp[hit[0]], calls + int(hit[0]) + 1, lg[hit[0]:hit[0] + 1], dg[hit[0]:hit[0] + 1]
            calls += len(grp)
        return None, calls, None, None

    def _rejection_flow_sample(self, init_samples, loglstar, enlargement_factor=1.1, constant_efficiency_factor=None,
                               cache=False, block=256):
        netG = self.trainer.netG

        def get_cache():
            z, log_det_J = netG.forward(init_samples)
            # envelope for the rejection step: max log|dx/dz| over the live points (sampler.py:556-561)
            self.max_log_det_J = enlargement_factor * float(torch.max(-log_det_J).item())
            self.max_r = float(np.max(np.linalg.norm(z.cpu().numpy(), axis=1)))

        if not cache or not hasattr(self, 'max_log_det_J'):
            get_cache()
        if constant_efficiency_factor is not None:
            enlargement_factor = (1 / constant_efficiency_factor) ** (1 / self.x_dim)
        ncall = 0
        while True:
            prior = getattr(netG, 'prior', None)
            if hasattr(prior, 'usample'):   # sampler.py:575-576: uniform box of the generalised-normal base
                z = np.asarray(prior.usample(sample_shape=(block,))).reshape(block, self.x_dim) * enlargement_factor
            else:                           # uniform in the ball of radius enlargement * max_r (sampler.py:579-583)
                z = np.random.randn(block, self.x_dim)
                r = enlargement_factor * self.max_r * np.random.rand(block) ** (1. / self.x_dim)
                z = z * (r / np.sqrt(np.sum(z ** 2, axis=1)))[:, None]
            rnd_u = np.random.rand(block)
            x, ld, logl, inbox = self._candidate_block(z)
            with np.errstate(over='ignore', invalid='ignore'):
                ratio = np.minimum(np.exp(ld - self.max_log_det_J), 1.0)
            # first test (sampler.py:593-597): Jacobian envelope; the second (sampler.py:599-604) zeroes the ratio
            # where logl is finite and below loglstar and repeats rnd_u < ratio
            passed = inbox & (rnd_u < ratio)
            j, calls, lj, dj = self._first_accepted(x, passed, logl, loglstar, strict=False)
            ncall += calls
            if j is not None:
                return x[j:j + 1], lj, dj, ncall
            block = min(4 * block, 65536)

    def _density_sample(self, loglstar, block=256):
        ncall = 0
        while True:
            z = self.trainer.get_prior_samples(block)
            x, ld, logl, inbox = self._candidate_block(z)
            j, calls, lj, dj = self._first_accepted(x, inbox, logl, loglstar, strict=True)
            ncall += calls
            if j is not None:
                return x[j:j + 1], lj, dj, ncall
            block = min(4 * block, 65536)

    # ---- chain files (sampler.py:494-527): getdist text format "weight -logL params..." ------------------------
    def _save_samples(self, samples, loglikes, weights=None, derived_samples=None, min_weight=1e-30, outfile='chain'):
        if self.logs is None:
            return
        if weights is None:
            weights = np.ones_like(loglikes)
        cols = [np.maximum(weights, min_weight)[:, None], -np.asarray(loglikes)[:, None], samples]
        if derived_samples is not None:
            cols.append(derived_samples)
        header = ''
        if self.param_names is not None:
            header = 'weight minusloglike ' + ' '.join(self.param_names)
        write_rows_e5(os.path.join(self.logs['chains'], outfile + '.txt'), np.concatenate(cols, axis=1), header=header)

    # ---- collectives -------------------------------------------------------------------------------------------------
    # One process per GPU, torch.distributed ('nccl' = RCCL over xGMI; 'gloo' on CPU for the tests).  A tensor argument stays
    # a tensor on the communication device (no host round trip under RCCL); a numpy argument comes back as numpy.
    def _comm_device(self):
        if torch.distributed.get_backend() == 'nccl':
            return torch.device('cuda', torch.cuda.current_device())
        return torch.device('cpu')

    def _all_gather_rows(self, arr):
        """concatenate equally-shaped per-rank arrays along axis 0 on every rank (C2)"""
        if not self.use_mpi:
            return arr
        as_numpy = not torch.is_tensor(arr)
        t = (torch.as_tensor(np.ascontiguousarray(arr)) if as_numpy else arr).to(self._comm_device()).contiguous()
        out = torch.empty((self.mpi_size * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        torch.distributed.all_gather_into_tensor(out, t)
        return out.cpu().numpy() if as_numpy else out

    def _broadcast(self, arr, src=0):
        if not self.use_mpi:
            return arr
        as_numpy = not torch.is_tensor(arr)
        t = (torch.as_tensor(np.ascontiguousarray(arr)) if as_numpy else arr).to(self._comm_device()).contiguous()
        torch.distributed.broadcast(t, src=src)
        return t.cpu().numpy() if as_numpy else t

    def _all_sum(self, value):
        if not self.use_mpi:
            return value
        t = torch.tensor([float(value)], dtype=torch.float64, device=self._comm_device())
        torch.distributed.all_reduce(t)
        return float(t.item())
