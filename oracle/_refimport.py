"""Import the upstream reference (read-only, /root/reference) inside THIS container only.

Test infrastructure. The reference needs two logging-only modules the image lacks
(tensorboard, getdist); neither touches arithmetic, so they are replaced
by inert placeholders in sys.modules before `import nnest`. Nothing here is imported by
the product package, bench.py or the -m gpu tests (the reference does not exist on the GPU box).
"""
import sys
import types

REFERENCE_ROOT = '/root/reference'


def import_reference():
    import matplotlib
    matplotlib.use('Agg')
    import torch
    import torch.utils  # noqa: F401

    if 'torch.utils.tensorboard' not in sys.modules:
        tb = types.ModuleType('torch.utils.tensorboard')

        class SummaryWriter(object):  # logging sink only
            def __init__(self, *a, **k):
                pass

            def add_scalar(self, *a, **k):
                pass

            def add_figure(self, *a, **k):
                pass

        tb.SummaryWriter = SummaryWriter
        sys.modules['torch.utils.tensorboard'] = tb
    if 'getdist' not in sys.modules:
        gd = types.ModuleType('getdist')
        gm = types.ModuleType('getdist.mcsamples')

        class MCSamples(object):
            pass

        gd.MCSamples = MCSamples
        gm.MCSamples = MCSamples
        gd.mcsamples = gm
        sys.modules['getdist'] = gd
        sys.modules['getdist.mcsamples'] = gm
    # on the path for the import only: the reference has a `tests` package of its own, and a path entry left behind would
    # shadow this repo's `tests` in every process spawned later (spawn hands the parent's sys.path to the child)
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        import nnest  # noqa: F401
        import nnest.nested, nnest.mcmc, nnest.sampler, nnest.trainer, nnest.networks  # noqa: F401,E401
    finally:
        sys.path.remove(REFERENCE_ROOT)
    return nnest
