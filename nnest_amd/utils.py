"""Host plumbing shared by the sampler front-ends: logger factory and the numbered run-directory tree
(same layout as the reference's nnest/utils/logger.py:9-22, :38-75 so its analysis scripts find the files)."""
import errno
import logging
import os
import sys


def create_logger(name, level=logging.INFO):
    logger = logging.getLogger(name)
    logger.setLevel(level)
    if not logger.handlers:
        handler = logging.StreamHandler(sys.stdout)
        handler.setFormatter(logging.Formatter('[%(name)s] [%(levelname)s] %(message)s'))
        logger.addHandler(handler)
        logger.propagate = False
    return logger


def _mkdir(path):
    try:
        os.makedirs(path)
    except OSError as e:
        if e.errno != errno.EEXIST:
            raise


def get_or_create_run_dir(run_dir, append_run_num=True):
    """<log_dir>/run<N>/{info,results,chains,checkpoint,plots}; an existing tree (has info/) is reused."""
    if os.path.isdir(os.path.join(run_dir, 'info')):
        created = False
    else:
        created = True
        _mkdir(run_dir)
        if append_run_num:
            run_num = sum(os.path.isdir(os.path.join(run_dir, i)) for i in os.listdir(run_dir)) + 1
            run_dir = os.path.join(run_dir, 'run%s' % run_num)
        for sub in ('', 'info', 'results', 'chains', 'checkpoint', 'plots'):
            _mkdir(os.path.join(run_dir, sub))
    return {'run_dir': run_dir, 'info': os.path.join(run_dir, 'info'), 'results': os.path.join(run_dir, 'results'),
            'chains': os.path.join(run_dir, 'chains'), 'checkpoint': os.path.join(run_dir, 'checkpoint'),
            'plots': os.path.join(run_dir, 'plots'), 'created': created}


class ScalarWriter(object):
    """Stands where the reference keeps a TensorBoard SummaryWriter (trainer.py:127-129; used by
    nested.py:467): scalars are appended to <path>/scalars.csv; figures are dropped."""

    def __init__(self, path=None):
        self.path = None if path is None else os.path.join(path, 'scalars.csv')
        self._buf = []

    def add_scalar(self, tag, value, step=None):
        if self.path is None:
            return
        self._buf.append('%s,%s,%r\n' % (tag, step, float(value)))
        if len(self._buf) >= 256:
            self.flush()

    def add_figure(self, *a, **k):
        pass

    def flush(self):
        if self.path is not None and self._buf:
            with open(self.path, 'a') as f:
                f.writelines(self._buf)
            self._buf = []

    def __del__(self):
        try:
            self.flush()
        except Exception:
            pass
