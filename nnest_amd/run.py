"""Command-line driver, the counterpart of the reference's examples/nested/run.py:14-89 (same flags where they
apply to this build).

  python -m nnest_amd.run --x_dim 50 --likelihood rosenbrock --num_live_points 1000 --mcmc_num_chains 1000
"""
import argparse
import datetime
import json
import os
import time

import numpy as np
import torch


def main(args):
    from nnest_amd.nested import NestedSampler
    from nnest_amd.likelihoods import Rosenbrock, Himmelblau, GaussianMix, Gaussian, Eggbox, GaussianShell

    name = args.likelihood.lower()
    if name == 'himmelblau':
        like, scale = Himmelblau(args.x_dim), 5.0
    elif name == 'rosenbrock':
        like, scale = Rosenbrock(args.x_dim), 5.0
    elif name == 'mixture':
        like, scale = GaussianMix(args.x_dim), 10.0
    elif name == 'gaussian':
        like, scale = Gaussian(args.x_dim, args.corr, lim=3), 3.0
    elif name == 'eggbox':
        like, scale = Eggbox(args.x_dim), 5 * np.pi
    elif name == 'shell':
        like, scale = GaussianShell(args.x_dim), 5.0
    else:
        raise ValueError('Likelihood not found')
    if args.seed >= 0:
        np.random.seed(args.seed)
        torch.manual_seed(args.seed)
    log_dir = os.path.join(args.log_dir, args.likelihood) + args.log_suffix
    base_dist = None
    if args.base_dist == 'gen_normal':   # examples/nested/run.py:20-21
        from .distributions import GeneralisedNormal
        base_dist = GeneralisedNormal(torch.zeros(args.x_dim), torch.ones(args.x_dim), torch.tensor(args.beta))
    sampler = NestedSampler(like.x_dim, like, transform=lambda x: scale * x, log_dir=log_dir,
                            num_live_points=args.num_live_points, hidden_dim=args.hidden_dim,
                            num_layers=args.num_layers, num_blocks=args.num_blocks, flow=args.flow, base_dist=base_dist,
                            scale=args.scale)
    start = time.time()
    sampler.run(train_iters=args.train_iters, mcmc_steps=args.mcmc_steps, volume_switch=args.switch, jitter=args.jitter,
                mcmc_num_chains=args.mcmc_num_chains, mcmc_dynamic_step_size=not args.mcmc_fixed_step_size,
                max_iters=args.max_iters)
    wall = time.time() - start
    print('Run time %s' % datetime.timedelta(seconds=wall))
    summary = {'likelihood': name, 'x_dim': like.x_dim, 'num_live_points': args.num_live_points,
               'mcmc_num_chains': args.mcmc_num_chains, 'logz': sampler.logz, 'logzerr': sampler.logzerr,
               'h': sampler.h, 'niter': sampler.niter, 'ncall': sampler.ncall, 'wall_s': wall,
               'retrains': sampler.num_retrains, 'batches': sampler.num_batches,
               'train_epochs_total': int(sampler.trainer.total_iters)}
    print(json.dumps(summary))
    with open(os.path.join(sampler.logs['results'], 'summary.json'), 'w') as f:
        json.dump(summary, f, indent=1)


if __name__ == '__main__':
    p = argparse.ArgumentParser()
    p.add_argument('--x_dim', type=int, default=2)
    p.add_argument('--train_iters', type=int, default=2000)
    p.add_argument('--mcmc_steps', type=int, default=0)
    p.add_argument('--mcmc_num_chains', type=int, default=10)
    p.add_argument('--num_live_points', type=int, default=1000)
    p.add_argument('-mcmc_fixed_step_size', action='store_true')
    p.add_argument('--switch', type=float, default=-1)
    p.add_argument('--hidden_dim', type=int, default=16)
    p.add_argument('--num_layers', type=int, default=1)
    p.add_argument('--flow', type=str, default='spline')
    p.add_argument('--num_blocks', type=int, default=3)
    p.add_argument('--jitter', type=float, default=-1)
    p.add_argument('--log_dir', type=str, default='logs')
    p.add_argument('--likelihood', type=str, default='rosenbrock')
    p.add_argument('--log_suffix', type=str, default='')
    p.add_argument('--max_iters', type=int, default=1000000)
    p.add_argument('--seed', type=int, default=-1)
    p.add_argument('--corr', type=float, default=0.99)
    p.add_argument('--base_dist', type=str, default='')
    p.add_argument('--beta', type=float, default=8.0)
    p.add_argument('--scale', type=str, default='')
    main(p.parse_args())
