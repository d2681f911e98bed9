"""ORACLE-SIDE GENERATOR -- TEST INFRASTRUCTURE ONLY (build container: imports /root/reference).

Runs the UNMODIFIED reference `nnest.nested.NestedSampler` (nnest/nested.py:97-510, `_mcmc_sample` sampler.py:229-463) with an
injected Trainer-shaped object (tests/oracle_trainer.OracleTrainer, the reference's own injection point sampler.py:50, :196-212)
on fixed numpy / torch seeds and records what the run reports: logz, ncall, niter, h, the dead-point chain and weights.
tests/test_reference_trajectory.py runs `nnest_amd.nested.NestedSampler` on the same trainer and seeds and asserts EQUALITY
(host protocol, CPU; on the GPU box against the committed values).

  python oracle/gen_nested_traj.py            -> tests/golden/nested_host_traj.json
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = {
    # name: (x_dim, likelihood, transform scale, live points, run keywords)
    'rosen2_mcmc': (2, 'Rosenbrock', 5.0, 60, dict(train_iters=30, mcmc_num_chains=5, max_iters=300, strategy=['mcmc'])),
    'rosen2_default': (2, 'Rosenbrock', 5.0, 60, dict(train_iters=30, mcmc_num_chains=5, max_iters=300)),
    'rosen2_fixed_step': (2, 'Rosenbrock', 5.0, 60, dict(train_iters=30, mcmc_num_chains=5, max_iters=300,
                                                         mcmc_dynamic_step_size=False)),
    'gmix4_default': (4, 'GaussianMix', 10.0, 80, dict(train_iters=20, mcmc_num_chains=8, max_iters=250)),
    'rosen2_converged': (2, 'Rosenbrock', 5.0, 40, dict(train_iters=20, mcmc_num_chains=4, mcmc_steps=6)),
}


def run_case(sampler_cls, like_cls, case, seed=0, **extra):
    from tests.oracle_trainer import OracleTrainer
    D, _, scale, N, kw = CASES[case]
    np.random.seed(seed)
    torch.manual_seed(seed)
    tr = OracleTrainer(D, seed=seed)
    s = sampler_cls(D, like_cls(D), transform=lambda x: scale * x, log_dir=tempfile.mkdtemp(dir='/tmp'), num_live_points=N,
                    flow='nvp', trainer=tr, log_level=40, **extra)
    s.run(**kw)
    with open(os.path.join(s.logs['results'], 'final.csv')) as f:   # niter, ncall, logz, logzerr, h as the run wrote them
        final = f.read().split()[1].split(',')
    return dict(logz=float(s.logz), niter=int(final[0]), ncall=int(final[1]), final_csv=final, total_calls=int(s.total_calls),
                num_trains=int(tr.num_trains), samples=np.asarray(s.samples, dtype=np.float64).tolist(),
                loglikes=np.asarray(s.loglikes, dtype=np.float64).tolist(),
                weights=np.asarray(s.weights, dtype=np.float64).tolist())


def main():
    import tests.oracle_trainer  # noqa: F401  (before /root/reference, which has a `tests` package too, goes on sys.path)
    from oracle._refimport import import_reference
    import_reference()
    from nnest.nested import NestedSampler as RefNestedSampler
    import nnest.likelihoods as ref_like
    doc = dict(what='reference NestedSampler (unmodified, /root/reference) driven with tests/oracle_trainer.OracleTrainer; '
                    'np.random.seed(0); torch.manual_seed(0); generator oracle/gen_nested_traj.py',
               cases={})
    for case, (D, like, scale, N, kw) in CASES.items():
        r = run_case(RefNestedSampler, getattr(ref_like, like), case)
        doc['cases'][case] = dict(x_dim=D, likelihood=like, scale=scale, num_live_points=N, run=kw, **r)
        print(case, r['logz'], r['ncall'], r['niter'], len(r['samples']))
    with open(os.path.join(ROOT, 'tests', 'golden', 'nested_host_traj.json'), 'w') as f:
        json.dump(doc, f)


if __name__ == '__main__':
    main()
