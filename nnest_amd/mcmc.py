"""MCMCSampler: the reference's flow-accelerated Metropolis front-end (nnest/mcmc.py:24-130) on this build's Sampler:
train the flow on a set of (normalised) training samples, then run `_mcmc_sample` with the likelihood and prior in the
proposal ratio (loglstar = None, sampler.py:371-410).  As in the reference, `mcmc_dynamic_step_size` is accepted and not
forwarded (mcmc.py:118-120): the chains run at the fixed step 2 / sqrt(x_dim).  Chain statistics (getdist ESS etc.,
sampler.py:474-492) are not computed here (out of scope)."""
import logging

import numpy as np

from .sampler import Sampler


class MCMCSampler(Sampler):

    def __init__(self, x_dim, loglike, prior=None, append_run_num=True, hidden_dim=16, num_slow=0, num_derived=0, batch_size=100,
                 flow='spline', num_blocks=3, num_layers=1, learning_rate=0.001, log_dir='logs/test', base_dist=None, scale='',
                 use_gpu=False, trainer=None, transform_prior=True, oversample_rate=-1, log_level=logging.INFO, param_names=None):
        super(MCMCSampler, self).__init__(x_dim, loglike, append_run_num=append_run_num, hidden_dim=hidden_dim, num_slow=num_slow,
                                          num_derived=num_derived, batch_size=batch_size, flow=flow, num_blocks=num_blocks,
                                          num_layers=num_layers, learning_rate=learning_rate, log_dir=log_dir, use_gpu=use_gpu,
                                          base_dist=base_dist, scale=scale, trainer=trainer, prior=prior,
                                          transform_prior=transform_prior, log_level=log_level, oversample_rate=oversample_rate,
                                          param_names=param_names)
        self.sampler = 'mcmc'

    def run(self, mcmc_steps, mcmc_num_chains, training_samples, mcmc_dynamic_step_size=True, stats_interval=100,
            output_interval=None, initial_jitter=0.01, final_jitter=0.01, init_samples=None):
        """mcmc.py:79-130"""
        mean = np.mean(training_samples, axis=0)
        std = np.std(training_samples, axis=0)
        training_samples = (training_samples - mean) / std          # normalise
        self.transform = lambda x: x * std + mean
        self._linear_scale = None                                    # the fused kernels only know x -> s * x
        self._fused_like_id = None
        self.trainer.train(training_samples, jitter=initial_jitter)
        samples, latent_samples, derived_samples, loglikes, scale, ncall = self._mcmc_sample(
            mcmc_steps, num_chains=mcmc_num_chains, stats_interval=stats_interval, output_interval=output_interval,
            init_samples=init_samples)   # mcmc.py:118-120 does not forward mcmc_dynamic_step_size: the chains keep a fixed step
        samples = self.transform(samples)
        self.samples = np.concatenate((samples, derived_samples), axis=2)
        self.latent_samples = latent_samples
        self.loglikes = loglikes
        self.logger.info('ncall: {:d}\n'.format(self.total_calls))
