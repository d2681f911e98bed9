"""nnest_amd: MI355X-native implementation of the nnest flow-transform + batched-proposal + likelihood
+ flow-training hot path (reference: adammoss/nnest v0.4.2).  See DESIGN.md / INTEGRATION.md."""
__version__ = '0.1.0'


def __getattr__(name):   # the reference's package-level names (nnest/__init__.py), imported on first use
    if name == 'NestedSampler':
        from .nested import NestedSampler
        return NestedSampler
    if name == 'MCMCSampler':
        from .mcmc import MCMCSampler
        return MCMCSampler
    if name == 'EnsembleSampler':
        raise NotImplementedError('EnsembleSampler (nnest/ensemble.py, an emcee front-end) is outside the scope of this build')
    raise AttributeError(name)
