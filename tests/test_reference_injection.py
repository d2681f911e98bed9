"""Protocol check at the reference's injection point (nnest/sampler.py:50, :196-212): the UNMODIFIED
reference NestedSampler is driven with a Trainer-shaped object exposing exactly the surface of
nnest_amd.Trainer(host_tensors=True) -- here the TEST-ONLY oracle-backed stand-in, since this container has no
GPU.  Runs only where /root/reference exists (the build container); skipped on the GPU box."""
import os

import numpy as np
import pytest
import torch

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'nnest')), reason='reference tree not present')


def test_reference_sampler_accepts_injected_trainer(tmp_path):
    from oracle._refimport import import_reference
    import_reference()
    from nnest.nested import NestedSampler as RefNestedSampler
    from nnest.likelihoods import Rosenbrock as RefRosenbrock
    from tests.oracle_trainer import OracleTrainer
    import nnest_amd.trainer as prod
    # the stand-in and the product expose the same protocol surface
    surface = ['forward', 'inverse', 'get_samples', 'get_latent_samples', 'get_prior_samples', 'train']
    for name in surface:
        assert callable(getattr(prod.Trainer, name)) and callable(getattr(OracleTrainer, name))
    np.random.seed(0)
    torch.manual_seed(0)
    tr = OracleTrainer(2, seed=0)
    s = RefNestedSampler(2, RefRosenbrock(2), transform=lambda x: 5 * x, log_dir=str(tmp_path), num_live_points=60,
                         flow='nvp', trainer=tr, log_level=40)
    assert s.trainer is tr
    s.run(train_iters=30, mcmc_num_chains=5, max_iters=120, strategy=['mcmc'])
    assert np.isfinite(s.logz) and tr.num_trains >= 1 and s.total_calls > 60
    assert s.samples.shape[1] == 2
