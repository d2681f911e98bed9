"""Trainer: drop-in for the reference's nnest.trainer.Trainer on the paths flow='nvp' and flow='spline', num_slow=0, N(0,I) base.

Same constructor keywords, attributes and method contracts as the reference (nnest/trainer.py:28-301), so
that `NestedSampler(..., trainer=Trainer(...))` -- the reference's own injection point (nnest/sampler.py:50,
:196-212) -- works unchanged.  All arithmetic runs in libnnest_hip.so on the GPU; there is no CPU fallback.
"""
import logging
import os
import time

import numpy as np
import torch

from . import _lib
from .flow import HipNVP, _as_dev_f32
from .utils import create_logger, ScalarWriter

EPOCH_CHUNK = 128  # epochs per training launch (bounds the device shuffle table; see include/nnest_hip.h)


class Trainer(object):
    best_validation_epoch = None
    best_validation_loss = None
    # train(..., rng_seed=s) is a pure function of (weights, optimiser state, samples, s) down to the last bit: the kernels
    # reduce in a fixed order (no atomics).  NestedSampler uses it to keep one replica per rank without broadcasting weights.
    replicable = True

    def __init__(self,
                 x_dim,
                 hidden_dim=16,
                 num_slow=0,
                 batch_size=100,
                 flow='spline',
                 scale='',
                 num_blocks=3,
                 num_layers=1,
                 base_dist=None,
                 load_model='',
                 log_dir='logs/test',
                 use_gpu=True,
                 log=True,
                 learning_rate=0.0001,
                 weight_decay=1e-6,
                 log_level=logging.INFO,
                 device=None,
                 seed=None,
                 host_tensors=False):
        if not torch.cuda.is_available():
            raise _lib.NnestHipError('nnest_amd.Trainer needs an MI355X (torch.cuda.is_available() is False); '
                                     'there is no CPU fallback')
        # 'maf' is not in the reference (trainer.py:83-100 knows choleksy, nvp, spline): build-defined, nnest_amd/maf.py
        if flow.lower() not in ('nvp', 'spline', 'choleksy', 'maf'):
            raise NotImplementedError('flow=%r (trainer.py:83-100 knows choleksy, nvp, spline; this build adds maf)' % flow)
        self.flow = flow.lower()
        if num_slow != 0:
            assert x_dim > num_slow                      # trainer.py:79
            if scale not in ('', None) or base_dist is not None:
                raise NotImplementedError('num_slow > 0 with a scale variant or a non-default base distribution')
        scale = '' if scale is None else scale
        if scale not in ('', 'translate', 'constant'):
            raise NotImplementedError("scale=%r: SingleSpeedNVP knows '', 'translate' and 'constant' (networks.py:330-332)" % scale)
        gen_normal = None
        if base_dist is not None:
            # N(0, I) -- the reference's default base (networks.py:51-57), which its notebooks also pass explicitly -- or
            # its GeneralisedNormal(0, 1, beta) (nnest/distributions/generalised_normal.py; run.py --base_dist gen_normal)
            if hasattr(base_dist, 'beta') and hasattr(base_dist, 'usample'):
                gen_normal = base_dist
            else:
                ok = isinstance(base_dist, torch.distributions.MultivariateNormal)
                if ok:
                    mean, cov = base_dist.mean.detach().cpu(), base_dist.covariance_matrix.detach().cpu()
                    ok = mean.shape == (x_dim,) and bool(torch.all(mean == 0)) and bool(torch.equal(cov, torch.eye(x_dim)))
                if not ok:
                    raise NotImplementedError('base_dist: N(0, I) or GeneralisedNormal(0, 1, beta)')
        if batch_size > 128 and (self.flow in ('maf', 'choleksy') or num_slow > 0):
            raise NotImplementedError("batch_size > 128 with flow=%r / num_slow > 0: only 'nvp' and 'spline' have the host-driven loop "
                                      "for minibatches beyond the training kernels' 128 row slots" % self.flow)
        # host_tensors=True: forward/inverse/... return CPU tensors and `.device` reads 'cpu' while the arithmetic
        # still runs on the GPU.  Needed only under the UNMODIFIED reference sampler, whose _mcmc_sample mixes CPU
        # tensors into the loop (nnest/sampler.py:305, :344) and therefore cannot consume CUDA tensors.
        self.gpu = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.host_tensors = bool(host_tensors)
        self.device = torch.device('cpu') if self.host_tensors else self.gpu
        self.x_dim = x_dim
        self.z_dim = x_dim
        self.batch_size = batch_size
        self.total_iters = 0
        self.num_slow = num_slow
        self.learning_rate = learning_rate
        self.weight_decay = weight_decay
        if self.flow == 'maf':
            if num_slow > 0 or scale != '':
                raise NotImplementedError("flow='maf' with num_slow > 0 or a scale variant")
            from .maf import HipMAF
            self.netG = HipMAF(x_dim, hidden_dim, num_blocks, num_layers, device=self.gpu, seed=seed)
        elif self.flow == 'choleksy':  # SingleSpeedCholeksy(x_dim)  (trainer.py:83-84; the reference ignores num_slow here too)
            from .cholesky import HipCholesky
            self.netG = HipCholesky(x_dim, device=self.gpu, seed=seed)
        elif self.flow == 'spline' and num_slow > 0:   # FastSlowSpline(num_fast, num_slow, hidden_dim, num_blocks)  (trainer.py:93-95)
            from .fastslow import HipFastSlowSpline
            self.netG = HipFastSlowSpline(x_dim - num_slow, num_slow, hidden_dim, num_blocks, device=self.gpu, seed=seed)
        elif self.flow == 'spline':  # SingleSpeedSpline(x_dim, hidden_dim, num_blocks, tail_bound=3)  (trainer.py:97-98)
            from .spline import HipSpline
            self.netG = HipSpline(x_dim, hidden_dim, num_blocks, num_bins=8, tail_bound=3.0, device=self.gpu, seed=seed)
        elif num_slow > 0:           # FastSlowNVP(num_fast, num_slow, hidden_dim, num_blocks, num_layers)  (trainer.py:86-88)
            from .fastslow import HipFastSlowNVP
            self.netG = HipFastSlowNVP(x_dim - num_slow, num_slow, hidden_dim, num_blocks, num_layers, device=self.gpu, seed=seed)
        else:
            self.netG = HipNVP(x_dim, hidden_dim, num_blocks, num_layers, device=self.gpu, seed=seed, scale=scale)
        self.replicable = self.flow in ('nvp', 'spline') and num_slow == 0   # the two single-launch training paths
        if batch_size > 128:
            self.netG.epoch_chunk = 1 << 30   # flow.train_epochs_host takes a run in one call
        if gen_normal is not None:
            self.netG.set_base(gen_normal)
            self.netG.prior = gen_normal
        if load_model:
            self.path = os.path.join(log_dir, load_model)
            sd = torch.load(os.path.join(self.path, 'models', 'netG.pt'))
            if self.flow == 'spline':
                # the 1x1 convs' permutations are not in the reference's state_dict (networks.py:634-635); this build
                # writes them beside it (netG_P.npz).  A loaded model counts as initialised (stated deviation: the
                # reference would re-run ActNorm's data-dependent init on the first forward batch and discard the
                # loaded s, t).
                pfile = os.path.join(self.path, 'models', 'netG_P.npz')
                P = None
                if os.path.exists(pfile):
                    P = dict(np.load(pfile))
                    P = P['P'] if 'P' in P else P
                self.netG.load_state_dict(sd, P)
                self.netG.data_dep_init_done = True
            else:
                self.netG.load_state_dict(sd)
        elif log_dir is not None:
            self.path = log_dir
            for sub in ('models', 'data', 'chains', 'plots'):
                os.makedirs(os.path.join(self.path, sub), exist_ok=True)
        else:
            self.path = None
        self.logger = create_logger(__name__, level=log_level)
        self.log = log
        self.writer = ScalarWriter(self.path)
        self.logger.info('Number of network params: [%s]' % (self.netG.reference_vector().size
                                                             if hasattr(self.netG, 'reference_vector') else self.netG.num_params))
        self.logger.info('Device [%s]' % self.device)

    # ------------------------------------------------------------------------------------------------
    def training_jitter(self, samples):
        """trainer.py:168-171 (jitter < 0): 0.2 * mean(cKDTree(samples).query(samples, 2) distances)"""
        return float(self._training_jitter_launch(samples).item())

    def _training_jitter_launch(self, samples):
        """the kernel of training_jitter queued, its result still on the device (train() reads it after it has queued the uploads
        of the training and validation rows: the read-back waits for work that is already done by then)"""
        x = torch.as_tensor(np.ascontiguousarray(samples, dtype=np.float64)).to(self.gpu)
        out = torch.zeros(1, dtype=torch.float64, device=self.gpu)
        with torch.cuda.device(self.gpu):
            _lib.check(_lib.load().nnest_training_jitter(_lib.ptr(x), x.shape[0], x.shape[1], _lib.ptr(out),
                                                         _lib.current_stream(self.gpu)))
        return out

    def train(self, samples, max_iters=10000, log_interval=100, save_interval=100, jitter=0.0,
              validation_fraction=0.1, patience=50, l2_norm=0.0, split=None, perms=None, noises=None, rng_seed=None):
        """Trainer.train (trainer.py:134-245).  `split`, `perms`, `noises` optionally replay recorded
        randomness (tests); by default the split comes from numpy's global RNG exactly as sklearn's
        train_test_split consumes it, the per-epoch shuffles from torch's CUDA generator and the jitter
        noise from the in-kernel Philox stream seeded from torch's CPU generator.  `rng_seed` (not in the reference) draws
        all three from that one integer instead, leaving the global generators untouched: ranks that pass the same seed
        train bit-identical replicas."""
        # l2_norm (trainer.py:395-399): loss += l2_norm * sum(param^2) after the reported loss is taken, i.e. the
        # gradient gains 2 * l2_norm * w -- the same term Adam's coupled weight decay adds (g += weight_decay * w)
        weight_decay = self.weight_decay + 2.0 * float(l2_norm)
        start_time = time.time()
        samples = np.asarray(samples)
        if self.path:
            if getattr(self, 'async_save', False):   # (a copy: the caller's live points move on while the worker writes)
                self._pending_files['originals'] = (np.array(samples), os.path.join(self.path, 'data', 'originals.npy'))
            else:
                np.save(os.path.join(self.path, 'data', 'originals.npy'), samples)
        jitter_dev = self._training_jitter_launch(samples) if jitter < 0 else None
        if self.log:
            self.logger.info('Number of training samples [%d]' % samples.shape[0])
        # train_test_split(samples, test_size=validation_fraction): ShuffleSplit draws rng.permutation(N) from
        # numpy's global state; test = first n_test, train = next n_train (sklearn/model_selection/_split.py)
        N = samples.shape[0]
        n_valid = int(np.ceil(validation_fraction * N))
        n_train = N - n_valid
        if n_train < 1 or n_valid < 1:
            raise ValueError('need at least one training and one validation sample (N=%d)' % N)
        gen = None
        if rng_seed is not None:
            rng_seed = int(rng_seed) & 0x7FFFFFFF
            gen = torch.Generator(device=self.gpu)
            gen.manual_seed(rng_seed)
        if split is not None:
            perm_split = np.asarray(split)
        elif rng_seed is not None:
            perm_split = np.random.RandomState(rng_seed).permutation(N)
        else:
            perm_split = np.random.permutation(N)
        x_valid = _as_dev_f32(samples[perm_split[:n_valid]], self.gpu)
        x_train = _as_dev_f32(samples[perm_split[n_valid:n_valid + n_train]], self.gpu)
        seed = rng_seed if rng_seed is not None else int(torch.empty((), dtype=torch.int64).random_().item())
        training_jitter = float(jitter_dev.item()) if jitter_dev is not None else jitter
        if self.log:
            self.logger.info('Training jitter [%5.4f]' % training_jitter)
        result, res, done, all_losses = None, None, 0, []
        while done < max_iters:
            chunk = min(getattr(self.netG, 'epoch_chunk', EPOCH_CHUNK), max_iters - done)
            if perms is not None:
                perm = torch.as_tensor(perms[done:done + chunk])
            else:  # DataLoader(shuffle=True): a fresh permutation per epoch (trainer.py:185)
                perm = torch.rand(chunk, n_train, device=self.gpu, generator=gen).argsort(dim=1, stable=True).int()
            nz = None if noises is None else torch.as_tensor(noises[done:done + chunk])
            res = self.netG.train_epochs(x_train, x_valid, perm, nz, seed=seed, jitter=training_jitter,
                                         batch=self.batch_size, max_epochs=chunk, patience=patience,
                                         lr=self.learning_rate, weight_decay=weight_decay, epoch_offset=done,
                                         resume=result is not None, finalize=(done + chunk >= max_iters), result=result)
            result = res['result']
            ran = res['epochs_run'] - done
            losses = res['losses'][:ran].cpu().numpy()
            all_losses.append(losses)
            if self.log:
                for k in range(ran):
                    epoch = done + k + 1
                    if epoch == 1 or epoch % log_interval == 0:
                        self.logger.info('Epoch [%i] train loss [%5.4f] validation loss [%5.4f]' % (epoch, losses[k, 0], losses[k, 1]))
            if ran > 0:   # (in bulk: 31 500 epochs of a config-2 run were 31 500 add_scalar calls)
                self.writer.add_scalars('loss', np.arange(self.total_iters + done + 1, self.total_iters + done + ran + 1), losses[:ran, 1])
            done = res['epochs_run']
            if res['stopped']:
                self.logger.info('Epoch [%i] ran out of patience' % done)
                break
        self.total_iters += done
        self.losses = np.concatenate(all_losses, 0) if all_losses else np.zeros((0, 2), np.float32)
        self.best_validation_epoch = res['best_epoch'] if res else 0
        self.best_validation_loss = res['best_validation_loss'] if res else float('inf')
        if self.path:
            # models/netG.pt after every train() (the reference writes it when patience runs out, trainer.py:226-230).  The state
            # dict is taken NOW; pickling and writing it (0.6 ms, 390 times in a config-2 run) happen on a worker thread beside
            # the next GPU work -- to a temporary file that then replaces netG.pt, so a reader never sees half a file.
            # (only where the owner asks for it -- NestedSampler.run does, and waits at its end; a plain train() call returns with
            # the file written, as in the reference)
            if getattr(self, 'async_save', False):
                # (the packed weights are read back NOW; the file follows at most SAVE_EVERY seconds later -- and at every checkpoint,
                # and at the end of the run: a config-2 run retrains 390 times in 4 s, and 390 pickles on the worker thread held the
                # interpreter lock for 0.2 s of the main thread's time)
                # (round 5: for the flows whose weights live in the library handle not even read back -- flush_pending_files() does
                # that when a file is actually due: 16 read-backs in a config-2 run instead of 392)
                self._pending_files['netG'] = (None if hasattr(self.netG, 'state_dict_from_packed') else self.netG.state_dict(),
                                               os.path.join(self.path, 'models', 'netG.pt'))
                if time.time() - getattr(self, '_last_flush', 0.0) >= self.SAVE_EVERY:
                    self.flush_pending_files()
            else:
                torch.save(self.netG.state_dict(), os.path.join(self.path, 'models', 'netG.pt'))
            if self.flow == 'spline':
                P = self.netG.P
                np.savez(os.path.join(self.path, 'models', 'netG_P.npz'), **(P if isinstance(P, dict) else {'P': P}))
        self.logger.info('Best epoch [%i] validation loss [%5.4f] train time (s) [%5.4f]]'
                         % (self.best_validation_epoch, self.best_validation_loss, time.time() - start_time))

    SAVE_EVERY = 0.25   # seconds between writes of models/netG.pt and data/originals.npy while a run owns the trainer (async_save)

    @property
    def _pending_files(self):
        return self.__dict__.setdefault('_pending', {})

    def flush_pending_files(self):
        """hand the latest models/netG.pt and data/originals.npy to the worker thread (slicing the packed weights into the state
        dict, pickling and writing happen there, to a temporary file that then replaces the old one: a reader never sees half
        a file).  Called by train() every SAVE_EVERY seconds, by the sampler in front of every checkpoint and at the end of run()."""
        pend, self._pending = self._pending_files, {}
        self._last_flush = time.time()
        netG = self.netG
        if 'netG' in pend and pend['netG'][0] is None:
            # the weights as they are NOW: nothing trains between the train() that left the entry and this call (train() itself, a
            # checkpoint or the end of the run, all on the thread that owns the device)
            pend['netG'] = (netG.store_packed(), pend['netG'][1])

        def work():
            if 'originals' in pend:
                np.save(pend['originals'][1], pend['originals'][0])
            if 'netG' in pend:
                state, path = pend['netG']
                tmp = path + '.tmp'
                torch.save(netG.state_dict_from_packed(state) if isinstance(state, np.ndarray) else state, tmp)
                os.replace(tmp, path)
        if pend:
            self.background_jobs().submit(work)

    def background_jobs(self):
        """the worker that writes files the run does not wait for (utils.BackgroundJobs), created at first use"""
        if getattr(self, '_jobs', None) is None:
            from .utils import BackgroundJobs
            self._jobs = BackgroundJobs()
        return self._jobs

    def wait_for_saves(self):
        """block until models/netG.pt holds the last trained weights (and everything else handed to background_jobs() is done)"""
        self.flush_pending_files()
        if getattr(self, '_jobs', None) is not None:
            self._jobs.wait()

    # ------------------------------------------------------------------------------------------------
    def forward(self, x, to_numpy=False):
        """trainer.py:247-257"""
        z, log_det_J = self.netG.forward(x)
        if to_numpy:
            return z.cpu().numpy(), log_det_J.cpu().numpy()
        return self._out(z), self._out(log_det_J)

    def inverse(self, z, to_numpy=False):
        """trainer.py:259-269"""
        x, log_det_J = self.netG.inverse(z)
        if to_numpy:
            return x.cpu().numpy(), log_det_J.cpu().numpy()
        return self._out(x), self._out(log_det_J)

    def get_prior_samples(self, num_samples, to_numpy=False):
        z = self.netG.prior_sample(num_samples)   # netG.prior.sample((n,))  trainer.py:272
        return z.cpu().numpy() if to_numpy else self._out(z)

    def _out(self, t):
        return t.cpu() if self.host_tensors else t

    def get_latent_samples(self, x, to_numpy=False):
        z, _ = self.forward(x, to_numpy=to_numpy)
        return z

    def get_samples(self, z, to_numpy=False):
        x, _ = self.inverse(z, to_numpy=to_numpy)
        return x

    def get_synthetic_samples(self, num_samples, to_numpy=False):
        x = self.netG.sample(num_samples)
        return x.cpu().numpy() if to_numpy else self._out(x)

    def log_probs(self, x, to_numpy=False):
        lp = self.netG.log_probs(x)
        return lp.cpu().numpy() if to_numpy else self._out(lp)
