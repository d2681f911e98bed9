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
import json
import logging
import os

import numpy as np
import torch

from .utils import create_logger, get_or_create_run_dir


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
                 mcmc_history=False):
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

        if transform is None:
            self.transform = lambda x: x
        else:
            def safe_transform(x):  # sampler.py:100-108
                if isinstance(x, list):
                    x = np.array(x)
                if len(x.shape) == 1:
                    assert x.shape[0] == self.x_dim
                    x = np.expand_dims(x, 0)
                return transform(x)
            self.transform = safe_transform
        self._linear_scale = detect_linear_scale(transform, x_dim)

        def safe_loglike(x):  # sampler.py:110-133
            if isinstance(x, list):
                x = np.array(x)
            if len(x.shape) == 1:
                assert x.shape[0] == self.x_dim
                x = np.expand_dims(x, 0)
            res = loglike(self.transform(x))
            self.total_calls += x.shape[0]
            if isinstance(res, tuple):
                logl, derived = res
            else:
                logl = res
                derived = np.empty((x.shape[0], 0))
            logl = np.array(logl, dtype=np.float64)  # float64 so that the -1e100 clamp below is representable
            if len(logl.shape) == 0:
                logl = np.expand_dims(logl, 0)
            logl[np.logical_not(np.isfinite(logl))] = -1e100
            if len(derived.shape) == 1:
                raise ValueError('Derived should have dimensions (batch size, num derived params)')
            if derived.shape[1] != self.num_derived:
                raise ValueError('Is the number of derived parameters correct?')
            return logl, derived
        self.loglike = safe_loglike

        sample_prior = getattr(prior, 'sample', None)
        self.sample_prior = sample_prior if callable(sample_prior) else None

        def safe_prior(x):  # sampler.py:137-163
            if isinstance(x, list):
                x = np.array(x)
            if len(x.shape) == 1:
                assert x.shape[0] == self.x_dim
                x = np.expand_dims(x, 0)
            if prior is None:
                return np.zeros(x.shape[0])
            if transform_prior:
                return np.array([prior(self.transform(r)) for r in x])
            return np.array([prior(r) for r in x])
        self.prior = safe_prior

        # process group (the reference: mpi4py, sampler.py:165-177)
        self.use_mpi = False
        self.mpi_size, self.mpi_rank = 1, 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.mpi_size = torch.distributed.get_world_size()
            self.mpi_rank = torch.distributed.get_rank()
            self.use_mpi = self.mpi_size > 1
        self.single_or_primary_process = (not self.use_mpi) or self.mpi_rank == 0

        if self.single_or_primary_process or (log_dir is not None and os.path.isdir(os.path.join(log_dir, 'info'))):
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
                                   log_dir=self.log_dir, log=self.single_or_primary_process, use_gpu=use_gpu,
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
        """64-bit seed for the in-kernel Philox streams, from torch's CPU generator (torch.manual_seed)."""
        return int(torch.empty((), dtype=torch.int64).random_().item())

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
                     seed=None):
        """Returns (samples, latent_samples, derived_samples, loglikes, scale, ncall) shaped as the reference
        (chain, step, dim), sampler.py:455-463.  On the fused path the step axis holds only the first and the
        last state unless the sampler was built with mcmc_history=True (nested.py:432-437 reads only those)."""
        if step_size <= 0.0:
            step_size = 2 / self.x_dim ** 0.5
        fused = (self._fused_like_id is not None and init_samples is not None and init_loglikes is not None
                 and prior_volume_steps == 1)   # loglstar None = the unconstrained branch, also in the kernel
        if fused:
            return self._mcmc_sample_fused(mcmc_steps, step_size, dynamic_step_size, init_samples, init_loglikes,
                                           loglstar, walker_offset, seed)
        return self._mcmc_sample_host(mcmc_steps, step_size, dynamic_step_size, num_chains, init_samples,
                                      init_loglikes, init_derived, loglstar, max_start_tries, prior_volume_steps)

    def _mcmc_sample_fused(self, mcmc_steps, step_size, dynamic, init_samples, init_loglikes, loglstar, walker_offset,
                           seed):
        netG = self.trainer.netG
        dev = netG.device
        C = init_samples.shape[0]
        z, _ = netG.forward(init_samples)                               # sampler.py:264
        logl = torch.as_tensor(np.ascontiguousarray(init_loglikes, dtype=np.float64)).to(dev)
        x0 = None
        if not self.mcmc_history:
            x0, _ = netG.inverse(z)                                    # sampler.py:266
        z0 = z.clone()
        res = netG.mh_steps(self._fused_like_id, self._linear_scale, z, logl, None if loglstar is None else float(loglstar), float(step_size),
                            int(mcmc_steps), dynamic=dynamic, seed=self._next_seed() if seed is None else seed,
                            walker_offset=walker_offset, history=self.mcmc_history, like_params=self._fused_like_params)
        ncall = int(res['n_call'].sum().item())
        nacc = int(res['n_accept'].sum().item())
        self.total_calls += ncall
        self.total_accepted += nacc
        self.total_rejected += C * int(mcmc_steps) - nacc
        if self.mcmc_history:
            samples = res['hist_x'].cpu().numpy()
            loglikes = res['hist_logl'].cpu().numpy()
            latent = np.stack([z0.cpu().numpy(), z.cpu().numpy()], axis=1)
        else:
            samples = torch.stack([x0, res['x']], dim=1).cpu().numpy()
            loglikes = np.stack([np.asarray(init_loglikes, dtype=np.float64), logl.cpu().numpy()], axis=1)
            latent = torch.stack([z0, z], dim=1).cpu().numpy()
        derived = np.empty((C, samples.shape[1], 0))
        scale = float(res['scale'].mean().item()) if dynamic else float(step_size)
        return samples, latent, derived, loglikes, scale, ncall

    def _mcmc_sample_host(self, mcmc_steps, step_size, dynamic, num_chains, init_samples, init_loglikes, init_derived,
                          loglstar, max_start_tries, prior_volume_steps):
        """The reference's step loop (sampler.py:246-463) with the flow passes on the GPU and the user's
        likelihood / prior callables on the host: the path for likelihoods the kernels do not know."""
        tr = self.trainer
        tr.netG.eval()
        samples, latent_samples, derived_samples, loglikes = [], [], [], []
        scale = step_size
        accept = reject = ncall = 0
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
        else:
            for i in range(max_start_tries):
                z = tr.get_prior_samples(num_chains)
                x = tr.get_samples(z, to_numpy=True)
                logl, derived = self.loglike(x)
                ncall += num_chains
                logl_prior = self.prior(x)
                if np.all(logl > -1e30) and np.all(logl_prior > -1e30):
                    break
                if i == max_start_tries - 1:
                    raise Exception('Could not find starting value')
        if not torch.is_tensor(z):
            z = torch.as_tensor(z)
        samples.append(x); latent_samples.append(z.cpu().numpy()); derived_samples.append(derived); loglikes.append(logl)
        for it in range(1, mcmc_steps + 1):
            x_t, log_det_J = tr.inverse(z)
            x = x_t.cpu().numpy()
            dz = torch.randn_like(z) * scale
            fast = False
            if self.num_slow > 0 and np.random.uniform() < self.oversample_rate:   # sampler.py:311-315, :378-382
                fast = True
                dz[:, 0:self.num_slow] = 0.0
            z_prime = z + dz
            x_prime_t, log_det_J_prime = tr.inverse(z_prime)
            x_prime = x_prime_t.cpu().numpy()
            log_ratio = (log_det_J_prime - log_det_J).cpu()
            logl_prior_prime = self.prior(x_prime)
            if loglstar is not None:
                log_ratio[torch.as_tensor(logl_prior_prime < -1e30)] = -np.inf
                rnd_u = torch.rand(log_ratio.shape)
                mask = (rnd_u < log_ratio.exp().clamp(max=1)).numpy().astype(bool)
                logl_prime = np.array(logl, dtype=np.float64, copy=True)
                derived_prime = np.copy(derived)
                idx = np.where(mask)[0]
                if len(idx) > 0:
                    lp, der = self.loglike(x_prime[idx])
                    ok = np.isfinite(lp) & (lp > loglstar)
                    ncall += len(idx)
                    if fast:
                        self.total_fast_calls += len(idx)
                    logl_prime[idx[ok]] = lp[ok]
                    derived_prime[idx[ok]] = der[ok]
                    mask[idx[~ok]] = False
            else:
                ncall += num_chains
                if fast:
                    self.total_fast_calls += num_chains
                logl_prime, derived_prime = self.loglike(x_prime)
                log_ratio = log_ratio + torch.as_tensor(logl_prime - logl) + torch.as_tensor(logl_prior_prime - logl_prior)
                rnd_u = torch.rand(log_ratio.shape)
                mask = (rnd_u < log_ratio.exp().clamp(max=1)).numpy().astype(bool)
            num_accepted = int(mask.sum())
            self.total_accepted += num_accepted
            self.total_rejected += num_chains - num_accepted
            if dynamic:
                if 2 * num_accepted > num_chains:
                    accept += 1
                else:
                    reject += 1
                if accept > reject:
                    scale *= np.exp(1. / (1 + accept))
                if accept < reject:
                    scale /= np.exp(1. / (1 + reject))
            logl = np.where(mask, logl_prime, logl)
            logl_prior = np.where(mask, logl_prior_prime, logl_prior)
            mt = torch.as_tensor(mask, device=z.device)[:, None]
            z = torch.where(mt, z_prime, z)
            x = np.where(mask[:, None], x_prime, x)
            derived = np.where(mask[:, None], derived_prime, derived)
            samples.append(x); latent_samples.append(z.cpu().numpy()); derived_samples.append(derived); loglikes.append(logl)
        samples = np.transpose(np.array(samples), axes=[1, 0, 2])
        latent_samples = np.transpose(np.array(latent_samples), axes=[1, 0, 2])
        derived_samples = np.transpose(np.array(derived_samples), axes=[1, 0, 2])
        loglikes = np.transpose(np.array(loglikes), axes=[1, 0])
        return samples, latent_samples, derived_samples, loglikes, scale, ncall

    # ---- prior rejection (sampler.py:529-543) ---------------------------------------------------------------
    def _rejection_prior_sample(self, loglstar, num_trials=None):
        if num_trials is None and self._fused_like_id is not None:
            # The reference draws one prior sample per likelihood call until one passes (sampler.py:531-538).  The
            # draws are independent, so the same rule is applied to a block per launch of the likelihood kernel: the
            # first candidate above loglstar is returned and ncall counts the candidates up to and including it.
            from . import flow
            ncall, block = 0, getattr(self, '_prior_block', 64)
            while True:
                x = self.sample_prior(block)
                logl = flow.loglike(self._fused_like_id, x, self._linear_scale, device=self.trainer.netG.device,
                                    like_params=self._fused_like_params).cpu().numpy()
                hit = np.where(logl > loglstar)[0]
                found = None
                for j in hit:   # the kernel works on float32(x); the stored value is the reference's float64 one
                    calls = self.total_calls
                    l64, d64 = self.loglike(x[j:j + 1])
                    self.total_calls = calls
                    if l64[0] > loglstar:
                        found = (int(j), l64, d64)
                        break
                if found is not None:
                    j, l64, d64 = found
                    self.total_calls += j + 1
                    # next block ~ 2 / (acceptance rate seen), bounded
                    self._prior_block = int(min(65536, max(64, 2 * block / max(1, len(hit)))))
                    return x[j:j + 1], l64, d64, ncall + j + 1
                self.total_calls += block
                ncall += block
                block = min(65536, 4 * block)
        if num_trials is None:
            ncall = 0
            while True:
                x = self.sample_prior(1)
                logl, derived = self.loglike(x)
                ncall += 1
                if logl > loglstar:
                    break
        else:
            x = self.sample_prior(num_trials)
            logl, derived = self.loglike(x)
            ncall = num_trials / np.sum(logl > loglstar)
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
            lg, dg = self.loglike(x[grp])
            good = lg > loglstar if strict else ~(np.isfinite(lg) & (lg < loglstar))
            hit = np.where(good)[0]
            if len(hit) > 0:
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
        np.savetxt(os.path.join(self.logs['chains'], outfile + '.txt'), np.concatenate(cols, axis=1), fmt='%.5E',
                   header=header, comments='#')

    # ---- collectives (host arrays) ---------------------------------------------------------------------------
    def _comm_device(self):
        return torch.device('cuda', torch.cuda.current_device()) if torch.distributed.get_backend() == 'nccl' else torch.device('cpu')

    def _all_gather_rows(self, arr):
        """concatenate equally-shaped per-rank arrays along axis 0 on every rank (C2)"""
        if not self.use_mpi:
            return arr
        t = torch.as_tensor(np.ascontiguousarray(arr)).to(self._comm_device())
        out = [torch.empty_like(t) for _ in range(self.mpi_size)]
        torch.distributed.all_gather(out, t)
        return torch.cat(out, dim=0).cpu().numpy()

    def _broadcast(self, arr, src=0):
        if not self.use_mpi:
            return arr
        t = torch.as_tensor(np.ascontiguousarray(arr)).to(self._comm_device())
        torch.distributed.broadcast(t, src=src)
        return t.cpu().numpy()

    def _all_sum(self, value):
        if not self.use_mpi:
            return value
        t = torch.tensor([float(value)], dtype=torch.float64, device=self._comm_device())
        torch.distributed.all_reduce(t)
        return float(t.item())
