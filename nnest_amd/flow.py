"""HipNVP: the RealNVP coupling stack resident on one MI355X, driven through the C ABI.

Mirrors SingleSpeedNVP / NormalizingFlowModel (reference nnest/networks.py:17-84, :328-347):
forward, inverse, log_probs, sample -- all on torch CUDA (ROCm) float32 tensors, computed by the HIP
kernels in libnnest_hip.so.  No PyTorch arithmetic is used for the flow itself.
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib


_STAGING = {}


def _staging_f32(n, device):
    """a pinned float32 staging buffer per (device, stream), grown on demand: callers on different streams (threads) never
    share one"""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    buf = _STAGING.get(key)
    if buf is None or buf.numel() < n:
        buf = torch.empty(max(int(n), 1 << 16), dtype=torch.float32).pin_memory()
        _STAGING[key] = buf
    return buf[:n]


def _as_dev_f32(x, device):
    """host array -> float32 device tensor [rows, D].  Host arrays are cast on the host (the same rounding as the device
    cast) with numpy and go through one pinned staging buffer; torch's own host-side cast/copy goes through its CPU thread
    pool, which on a many-core host measured 3-6 ms per call inside a run (tools/run_timing.py) against 0.1 ms here."""
    device = torch.device(device)
    if torch.is_tensor(x) and x.device.type == 'cpu' and device.type == 'cuda':
        x = x.detach().numpy()
    if not torch.is_tensor(x) and device.type == 'cuda':
        a = np.asarray(x)
        if a.ndim == 1:
            a = a[None, :]
        out = torch.empty(a.shape, dtype=torch.float32, device=device)
        if a.size:
            stage = _staging_f32(a.size, device)
            # cast and copy in ONE pass, straight into the pinned buffer (numpy, not Tensor.copy_: torch's CPU copy wakes its whole
            # thread pool -- 5.7 ms per call on a 256-core host, tools/run_timing.py)
            np.copyto(stage.numpy().reshape(a.shape), a, casting='unsafe')
            out.view(-1).copy_(stage, non_blocking=True)
            torch.cuda.current_stream(device).synchronize()  # the staging buffer is free for the next call
        return out
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    if not torch.is_tensor(x):
        x = torch.as_tensor(x)
    x = x.to(device=device, dtype=torch.float32)
    if x.dim() == 1:
        x = x[None, :]
    return x.contiguous()


SCALE_MODES = {'': 0, 'translate': 1, 'constant': 2}
TRAIN_KERNEL_MAX_BATCH = 128   # row slots of the one-launch training kernels (nnest_train.hip TRAIN_MAX_ROWS; nnest_spline_train.hip)


class _HipFlow(object):
    """What the two flow families share: the pass / proposal entry points differ only in the C symbol
    (`self._sym[...]`, set by the subclass) -- same tensors in, same tensors out."""

    base_beta = 0.0   # 0: N(0, I); > 0: GeneralisedNormal(0, 1, beta)
    base_dist = None  # the distribution object handed to Trainer(base_dist=...), if any

    def set_base(self, base_dist):
        """NormalizingFlowModel(prior=...) (networks.py:47-59): None / MultivariateNormal(0, I) or GeneralisedNormal(0, 1, beta)"""
        beta = 0.0
        if base_dist is not None and hasattr(base_dist, 'beta'):
            loc = torch.as_tensor(base_dist.loc, dtype=torch.float32).detach().cpu()
            scale = torch.as_tensor(base_dist.scale, dtype=torch.float32).detach().cpu()
            if not (bool(torch.all(loc == 0)) and bool(torch.all(scale == 1))):
                raise NotImplementedError('GeneralisedNormal base: only loc = 0, scale = 1 (what the reference constructs)')
            beta = float(base_dist.beta.item()) if torch.is_tensor(base_dist.beta) else float(base_dist.beta)
            if not beta > 0:
                raise ValueError('beta must be > 0')
            self.base_dist = base_dist
        with torch.cuda.device(self.device):
            _lib.check(self._sym['set_base'](self._h, ctypes.c_float(beta)))
        self.base_beta = beta

    def prior_sample(self, num_samples):
        """self.prior.sample((n,)) (networks.py:80): N(0, I) draws, or scipy gennorm draws for the generalised normal"""
        if self.base_dist is not None:
            return _as_dev_f32(self.base_dist.sample((int(num_samples),)).reshape(int(num_samples), self.D), self.device)
        return torch.randn(int(num_samples), self.D, device=self.device)

    def eval(self):
        return self

    def train(self, mode=True):
        return self

    def parameters(self):
        return list(self.state_dict().values())

    def _pass(self, fn, x, want_logdet=True):
        x = _as_dev_f32(x, self.device)
        N = x.shape[0]
        if x.shape[1] != self.D:
            raise ValueError('expected [N, %d], got %s' % (self.D, tuple(x.shape)))
        out = torch.empty_like(x)
        ld = torch.empty(N, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(fn(self._h, _lib.ptr(x), _lib.ptr(out), _lib.ptr(ld), N, _lib.current_stream(self.device)))
        return out, ld

    def forward(self, x):
        """NormalizingFlow.forward (networks.py:24-32): (z, log_det)"""
        return self._pass(self._sym['forward'], x)

    def inverse(self, z):
        """NormalizingFlow.inverse (networks.py:34-42): (x, log_det)"""
        return self._pass(self._sym['inverse'], z)

    def log_probs(self, x):
        """NormalizingFlowModel.log_probs (networks.py:71-76)"""
        x = _as_dev_f32(x, self.device)
        out = torch.empty(x.shape[0], dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._sym['log_probs'](self._h, _lib.ptr(x), _lib.ptr(out), x.shape[0],
                                              _lib.current_stream(self.device)))
        return out

    def sample(self, num_samples=None, noise=None):
        """NormalizingFlowModel.sample (networks.py:78-84)"""
        if noise is None:
            noise = self.prior_sample(num_samples)
        x, _ = self.inverse(noise)
        return x

    def inverse_loglike(self, like_id, like_scale, z, want_x=True, like_params=None):
        """K3: x = f^-1(z), logdet, box-prior flag, logl = loglike(like_scale * x) in one launch."""
        z = _as_dev_f32(z, self.device)
        N = z.shape[0]
        x = torch.empty_like(z) if want_x else None
        ld = torch.empty(N, dtype=torch.float32, device=self.device)
        logl = torch.empty(N, dtype=torch.float64, device=self.device)
        inbox = torch.empty(N, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            lk = _lib.like_spec(like_id, like_scale, like_params)
            _lib.check(self._sym['inverse_loglike'](self._h, ctypes.byref(lk), _lib.ptr(z),
                                                    _lib.ptr(x), _lib.ptr(ld), _lib.ptr(logl), _lib.ptr(inbox), N,
                                                    _lib.current_stream(self.device)))
        return x, ld, logl, inbox

    def mh_steps(self, like_id, like_scale, z, logl, loglstar, step_size, steps, dynamic=False, noise=None, seed=0,
                 walker_offset=0, history=False, like_params=None, lag=None, form=None, warm=None):
        """K4: `steps` constrained Metropolis steps for all walkers in one launch (Sampler._mcmc_sample,
        sampler.py:229-463).  z [C,D] float32 and logl [C] float64 are updated in place.
        noise = (dz [steps,C,D], u [steps,C]) replays recorded draws; None = in-kernel Philox.
        loglstar None / NaN = the unconstrained branch (sampler.py:371-410): likelihood and box prior in the ratio.
        dynamic: False | True or 'batch' (sampler.py:422-431 over all C walkers; `lag` steps between a step and the
        scale that reflects its count, 0 = the reference exactly) | 'group' (per 16 walkers, shard-invariant).
        warm: with a lagged batch rule, the first `warm` steps apply it exactly (NNEST_MH_WARM); None = default_warm().
        form: None (by population) | 'solo' | 'quad' | 'team' | 'reg' | 'image' (include/nnest_hip.h NNEST_MH_FORM_*)."""
        free = loglstar is None or loglstar != loglstar
        loglstar = 0.0 if free else loglstar
        assert z.is_cuda and z.dtype == torch.float32 and z.is_contiguous()
        assert logl.is_cuda and logl.dtype == torch.float64 and logl.is_contiguous()
        C = z.shape[0]
        dev = self.device
        x = torch.empty_like(z)
        n_acc = torch.empty(C, dtype=torch.int32, device=dev)    # (every form writes both counters of every walker: no fill launches
        n_call = torch.empty(C, dtype=torch.int32, device=dev)   # in front of the kernel)
        ngroups = (C + 15) // 16
        scale_out = torch.empty(max(ngroups, 1), dtype=torch.float32, device=dev)
        hx = torch.empty(C, steps + 1, self.D, dtype=torch.float32, device=dev) if history else None
        hl = torch.empty(C, steps + 1, dtype=torch.float64, device=dev) if history else None
        dz = u = None
        if noise is not None:
            dz = _as_dev_f32(noise[0].reshape(-1, self.D), dev)
            u = noise[1].to(device=dev, dtype=torch.float32).contiguous()
            assert dz.shape[0] == steps * C and u.numel() == steps * C
        if lag is None and dynamic in (True, 'batch'):
            lag = self.default_lag(C, form)
        if warm is None:
            warm = self.default_warm(C, dynamic, lag, form)
        flags = _lib.mh_flags(dynamic, free, lag, form, warm)
        sync = None
        if flags & _lib.MH_DYNAMIC_BATCH:   # per-step batch counters, zero at every launch; last word = error flag
            # a double buffer per step count, zeroed ONCE: each launch zeroes the half the next launch will use (in-kernel where the
            # solo form runs: the fill launch in front of every K4 launch -- 3 % of a config-2 launch -- is gone)
            W = int(self._lib.nnest_mh_sync_words(int(steps)))
            pool = self.__dict__.setdefault('_sync_pool', {})
            ent = pool.get(int(steps))
            if ent is None:
                ent = pool[int(steps)] = [torch.zeros(2 * W, dtype=torch.int64, device=dev), 0]
            half = ent[1]
            sync = ent[0][half * W:(half + 1) * W]
            flags |= _lib.MH_SYNC_ZERO_PREV if half else _lib.MH_SYNC_ZERO_NEXT
        with torch.cuda.device(dev):
            lk = _lib.like_spec(like_id, like_scale, like_params)
            try:
                _lib.check(self._sym['mh'](
                    self._h, ctypes.byref(lk), _lib.ptr(z), _lib.ptr(x), _lib.ptr(logl), float(loglstar),
                    float(step_size), int(steps), C, flags, _lib.ptr(dz), _lib.ptr(u),
                    int(seed) & 0xFFFFFFFFFFFFFFFF, int(walker_offset), _lib.ptr(hx), _lib.ptr(hl), _lib.ptr(n_acc),
                    _lib.ptr(n_call), _lib.ptr(scale_out), _lib.ptr(sync), _lib.current_stream(dev)))
            except Exception:
                # a refused launch ran nothing: the half it was given may or may not be the one the last launch zeroed for it,
                # and the other half was not zeroed for the launch after -- start the pair over from fresh zeros
                if sync is not None:
                    self._sync_pool.pop(int(steps), None)
                raise
        if sync is not None:
            ent[1] ^= 1   # the halves change roles only behind a launch that ran (it zeroed the other one)
        # the kernels report in the count's bit 30 whether EVERY coordinate of the chain's last x differs from its first
        # (include/nnest_hip.h NNEST_MH_ALL_MOVED): the reference's usable-chain test, nested.py:432
        # (split on first use: two element-wise launches that a caller who reads neither does not pay for)
        return _MhResult(x=x, n_accept_word=n_acc, n_call=n_call, scale=scale_out, hist_x=hx, hist_logl=hl, sync=sync)

    def slice_steps(self, like_id, like_scale, z, logl, loglstar, width, steps, max_stepout=8, max_shrink=32, noise=None, seed=0,
                    walker_offset=0, history=False, like_params=None):
        """SLICE proposal in latent space (nnest_slice_steps; BUILD-DEFINED: the reference has none, nnest/sampler.py:310-316 is
        random-walk Metropolis): `steps` slice-sampling updates (stepping out + shrinkage along a random direction) of every walker
        under the hard constraint logL > loglstar, the target the reference's Metropolis step leaves invariant.  z [C,D] float32 and
        logl [C] float64 are updated in place.  noise = dz [steps,C,D] replays recorded directions.  Returns x, n_call (candidates
        whose likelihood decided), n_move, moved (nested.py:432), n_eval (flow evaluations), hist_x."""
        assert z.is_cuda and z.dtype == torch.float32 and z.is_contiguous()
        assert logl.is_cuda and logl.dtype == torch.float64 and logl.is_contiguous()
        C, dev = z.shape[0], self.device
        x = torch.empty_like(z)
        n_call = torch.empty(C, dtype=torch.int32, device=dev)
        n_move = torch.empty(C, dtype=torch.int32, device=dev)
        n_eval = torch.empty(C, dtype=torch.int32, device=dev)
        hx = torch.empty(C, steps + 1, self.D, dtype=torch.float32, device=dev) if history else None
        dz = None
        if noise is not None:
            dz = _as_dev_f32(noise.reshape(-1, self.D), dev)
            assert dz.shape[0] == steps * C
        with torch.cuda.device(dev):
            lk = _lib.like_spec(like_id, like_scale, like_params)
            _lib.check(self._lib.nnest_slice_steps(self._h, ctypes.byref(lk), _lib.ptr(z), _lib.ptr(x), _lib.ptr(logl), float(loglstar),
                                                   float(width), int(steps), C, int(max_stepout), int(max_shrink), _lib.ptr(dz),
                                                   int(seed) & 0xFFFFFFFFFFFFFFFF, int(walker_offset), _lib.ptr(hx), _lib.ptr(n_call),
                                                   _lib.ptr(n_move), _lib.ptr(n_eval), _lib.current_stream(dev)))
        return dict(x=x, n_call=n_call, n_move=n_move & (_lib.MH_ALL_MOVED - 1), moved=(n_move & _lib.MH_ALL_MOVED) != 0, n_eval=n_eval,
                    hist_x=hx)

    def fill_slice_noise(self, steps, C, seed=0, walker_offset=0):
        """the directions nnest_slice_steps draws (Philox normals), exported for the checker: dz [steps, C, D]"""
        dz = torch.empty(steps, C, self.D, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.nnest_slice_fill_noise(_lib.ptr(dz), steps, C, self.D, int(seed) & 0xFFFFFFFFFFFFFFFF,
                                                        int(walker_offset), _lib.current_stream(self.device)))
        return dz

    def mh_form_for(self, C, dynamic=False, lag=None, free=False, form=None, warm=0):
        """the K4 form (name) `mh_steps` runs for C walkers under this step rule -- asked of the library
        (nnest_mh_form_for), which knows the shapes each form is built for; None if the launch would be refused.  A caller
        that shards one batch over ranks asks for the WHOLE batch and pins the answer on every shard."""
        fn = getattr(self._lib, 'nnest_mh_form_for', None) if self._sym.get('mh') is getattr(self._lib, 'nnest_mh_constrained_steps', None) else None
        if fn is None:
            return None
        f = fn(self._h, int(C), _lib.mh_flags(dynamic, free, lag, form, warm))
        return _lib.MH_FORM_NAMES.get(f)

    def default_lag(self, C, form=None):
        """steps between an MCMC step and the proposal scale that reflects its batch-wide accept count when the caller names
        none.  The count of a step reaches every compute unit about 4.5 us after the step ended (atomics -> publishing wave ->
        one fetch); the lag is what keeps that off the step's critical path: 4 steps of the 2 us forms, 8 steps of the solo
        form's 1.2 us.  0 is the reference's rule exactly."""
        if form == 'solo' or (form is None and self.mh_form_for(C, dynamic='batch', lag=_lib.MH_SOLO_LAG) == 'solo'):
            return _lib.MH_SOLO_LAG
        return _lib.MH_DEFAULT_LAG

    def default_warm(self, C, dynamic, lag, form=None):
        """exact steps in front of a lagged batch rule when the caller names none: MH_WARM_STEPS where the form that runs
        implements them (the solo form), 0 elsewhere.  Every launch restarts the rule from the caller's step_size with gain
        1 / (1 + votes): its first votes move the scale by e-folds, and a lag there is so many steps spent at the initial
        scale."""
        if dynamic not in (True, 'batch') or not lag:
            return 0
        w = _lib.MH_WARM_STEPS
        return w if self.mh_form_for(C, dynamic='batch', lag=lag, form=form, warm=w) == 'solo' else 0

    @staticmethod
    def check_sync(res):
        """raise if a bounded wait of the batch-wide step rule ran out (a workgroup was not resident); synchronises"""
        if res.get('sync') is not None and int(res['sync'][-1].item()) != 0:
            raise _lib.NnestHipError('batch-wide step rule: a wait on the per-step counters ran out')

    def fill_noise(self, steps, C, seed=0, walker_offset=0):
        dz = torch.empty(steps, C, self.D, dtype=torch.float32, device=self.device)
        u = torch.empty(steps, C, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.nnest_mh_fill_noise(_lib.ptr(dz), _lib.ptr(u), steps, C, self.D,
                                                     int(seed) & 0xFFFFFFFFFFFFFFFF, int(walker_offset),
                                                     _lib.current_stream(self.device)))
        return dz, u


class _MhResult(dict):
    """what mh_steps returns; 'n_accept' (the count) and 'moved' (the reference's usable-chain test, nested.py:432) are split off
    the kernel's word (NNEST_MH_ALL_MOVED) when first asked for"""

    def __missing__(self, key):
        if key in ('n_accept', 'moved'):
            w = dict.__getitem__(self, 'n_accept_word')
            self['moved'] = (w & _lib.MH_ALL_MOVED) != 0
            self['n_accept'] = w & (_lib.MH_ALL_MOVED - 1)
            return dict.__getitem__(self, key)
        raise KeyError(key)

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default


def train_epochs_host(net, xtrain, xvalid, perm, noise, seed, jitter, batch, max_epochs, patience, lr, weight_decay, chunk_rows):
    """Trainer.train's epoch loop (nnest/trainer.py:198-241, :384-418) driven from the host for the shapes the one-launch training
    kernels do not take (batch_size > 128: they hold a minibatch in one grid of 128 row slots): per minibatch the gradient of
    -mean(log_probs) from `net.loss_grad` over chunks of at most `chunk_rows` rows -- the mean over M rows is the chunk means
    weighted m_c / M, exact up to rounding -- and one `net.adam_step`; the early-stopping books as the reference keeps them.  A slower
    path (a few launches per minibatch), not a refusal.  Arguments and return value as HipNVP.train_epochs."""
    dev = net.device
    xtrain = _as_dev_f32(xtrain, dev)
    xvalid = _as_dev_f32(xvalid, dev)
    n_train, n_valid, D = xtrain.shape[0], xvalid.shape[0], xtrain.shape[1]
    perm = perm.to(device=dev, dtype=torch.int64).view(max_epochs, n_train)
    if noise is not None:
        noise = noise.to(device=dev, dtype=torch.float32).view(max_epochs, n_train, D)
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
    losses = np.zeros((max(max_epochs, 1), 2), np.float32)
    best, best_epoch, counter, stopped, epochs_run = float('inf'), 0, 0, False, 0
    best_w = net.store_packed()                                            # best_model = deepcopy(netG)  trainer.py:194
    for epoch in range(max_epochs):
        rows_all = xtrain[perm[epoch]]
        if jitter != 0.0:                                                  # data + jitter * randn_like(data)  trainer.py:392
            nz = noise[epoch] if noise is not None else torch.randn(rows_all.shape, device=dev, generator=gen)
            rows_all = rows_all + float(jitter) * nz
        tot = torch.zeros((), dtype=torch.float32, device=dev)
        for lo in range(0, n_train, int(batch)):
            rows = rows_all[lo:lo + int(batch)]
            M = rows.shape[0]
            if getattr(net, 'data_dep_init_done', True) is False:          # ActNorm: the first minibatch pushed forward through a
                net.actnorm_init(rows.contiguous())                        # fresh spline flow initialises it (networks.py:698-705)
            grad = loss = None
            for c0 in range(0, M, int(chunk_rows)):
                part = rows[c0:c0 + int(chunk_rows)].contiguous()
                l_c, g_c = net.loss_grad(part)
                wgt = part.shape[0] / M
                grad = g_c * wgt if grad is None else grad + g_c * wgt
                loss = l_c[0] * wgt if loss is None else loss + l_c[0] * wgt
            net.adam_step(grad, lr, weight_decay)                          # trainer.py:400-401
            tot = tot + loss                                               # train_loss += loss.item()  trainer.py:398
        vsum = -net.log_probs(xvalid).mean()                               # one full batch  trainer.py:190, :405-418
        both = torch.stack([tot, vsum.to(tot.dtype)]).cpu()                # one read-back per epoch
        train_loss, valid_loss = float(both[0]) / n_train, float(both[1]) / n_valid
        losses[epoch] = (train_loss, valid_loss)
        epochs_run = epoch + 1
        if valid_loss < best:                                              # trainer.py:205-209
            best, best_epoch, counter, best_w = valid_loss, epoch + 1, 0, net.store_packed()
        counter += 1
        if counter > patience:                                             # trainer.py:223-232
            stopped = True
            break
    if hasattr(net, 'P'):
        net.load_packed(best_w, net.P)                                     # netG.load_state_dict(best_model)  trainer.py:241
    else:
        net.load_packed(best_w)
    return dict(losses=torch.from_numpy(losses), epochs_run=epochs_run, best_epoch=best_epoch, best_validation_loss=best,
                last_train_loss=float(losses[max(epochs_run - 1, 0), 0]), counter=counter, stopped=stopped, result=None)


def native_hidden(H):
    """the hidden width the kernels are instantiated for that holds H: 16, 32 or 64 (one, two or four 16-wide matrix-core tiles);
    beyond 64 the next multiple of 16 (the library says what it cannot take)"""
    for w in (16, 32, 64):
        if H <= w:
            return w
    return -(-H // 16) * 16


def pad_index(blocks):
    """blocks: [(user_shape, native_shape, native_offset)] in the order of the user's packed vector -> for every element of that
    vector its position in the native one (a tensor of user_shape sits in the leading corner of a zero tensor of native_shape)."""
    out = []
    for su, sn, off in blocks:
        if len(su) == 0:
            out.append(np.array([off], dtype=np.int64))
            continue
        ix = np.indices(su).reshape(len(su), -1)
        out.append(np.ravel_multi_index(ix, sn).astype(np.int64) + off)
    return np.concatenate(out) if out else np.zeros(0, np.int64)


class _PaddedVectors(object):
    """A flow whose hidden width is not a multiple of the 16-wide matrix-core tile runs on a native handle of the next multiple:
    the extra hidden units have zero weights in and out and zero biases.  That is EXACT, not an approximation: their
    pre-activations are 0, tanh 0 = relu 0 = leaky_relu 0 = 0, nothing flows out of them; in reverse mode the gradient reaching them
    is W_out^T g = 0 and the gradients of their own weights are products with those zeros; Adam with coupled weight decay maps
    (w, g, m, v) = (0, 0, 0, 0) to itself.  (The reference accepts any hidden_dim: nnest/networks.py:253-287, trainer.py:32-48.)
    `_pidx` (None when nothing is padded): position of every element of the user's packed vector in the native one."""
    _pidx = None
    _pidx_dev = None
    _native_params = None

    def _set_pad_index(self, idx, native_params):
        self._native_params = int(native_params)
        if idx is None or (len(idx) == native_params and np.array_equal(idx, np.arange(native_params))):
            self._pidx = self._pidx_dev = None
            return
        self._pidx = idx
        self._pidx_dev = torch.from_numpy(idx).to(self.device)

    def _to_native(self, vec):
        vec = np.ascontiguousarray(vec, dtype=np.float32)
        if self._pidx is None:
            return vec
        out = np.zeros(self._native_params, np.float32)
        out[self._pidx] = vec
        return out

    def _from_native(self, vec):
        return vec if self._pidx is None else np.ascontiguousarray(vec[self._pidx])

    def _native_buffer(self):
        return np.empty(self._native_params if self._native_params is not None else self.num_params, np.float32)

    def _to_native_dev(self, t):
        if self._pidx_dev is None:
            return t.contiguous()
        out = torch.zeros(self._native_params, dtype=torch.float32, device=self.device)
        out[self._pidx_dev] = t
        return out

    def _from_native_dev(self, t):
        return t if self._pidx_dev is None else t[self._pidx_dev].contiguous()


class HipNVP(_PaddedVectors, _HipFlow):
    """num_inputs=D, num_hidden=H, num_blocks=B, num_layers=L, scale as SingleSpeedNVP (networks.py:328-347).

    scale='translate' / 'constant' (translate-only couplings; 'constant' adds a ScaleLayer scalar after each block):
    state_dict() has the reference's keys (no scale nets; `flow.flows.<2b+1>.scale` scalars); the packed vector of
    the C ABI keeps the scale_net slots (zero) and appends the scalars (include/nnest_hip.h)."""

    def __init__(self, num_inputs, num_hidden=16, num_blocks=3, num_layers=1, device=None, seed=None, scale=''):
        if not torch.cuda.is_available():
            raise _lib.NnestHipError('HipNVP needs an MI355X visible to PyTorch-ROCm (torch.cuda.is_available() is False); '
                                     'there is no CPU fallback')
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.D, self.H, self.B, self.L = int(num_inputs), int(num_hidden), int(num_blocks), int(num_layers)
        self.num_inputs = self.D
        if scale not in SCALE_MODES:
            raise ValueError("scale=%r: expected '', 'translate' or 'constant' (networks.py:330-332)" % (scale,))
        self.scale = scale
        self._lib = _lib.load()
        L = self._lib
        self._sym = dict(forward=L.nnest_nvp_forward, inverse=L.nnest_nvp_inverse, log_probs=L.nnest_nvp_log_probs,
                         inverse_loglike=L.nnest_nvp_inverse_loglike, mh=L.nnest_mh_constrained_steps, set_base=L.nnest_nvp_set_base)
        self._h = ctypes.c_void_p()
        self._Hn = native_hidden(self.H)     # the native handle's hidden width (_PaddedVectors)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.nnest_nvp_create_scaled(self.D, self._Hn, self.B, self.L, SCALE_MODES[scale],
                                                         ctypes.byref(self._h)))
        self._init_padding()
        self.prior = torch.distributions.MultivariateNormal(torch.zeros(self.D, device=self.device),
                                                            torch.eye(self.D, device=self.device))
        self.load_packed(self.default_init(seed))

    def __del__(self):
        try:
            if getattr(self, '_h', None) is not None and self._h.value:
                self._lib.nnest_nvp_destroy(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass

    def _init_padding(self):
        """num_params and the packed layout are the USER's (hidden width H); the native vector has width _Hn"""
        D, H, Hn, L, B = self.D, self.H, getattr(self, '_Hn', self.H), self.L, self.B
        native = self._lib.nnest_nvp_num_params(self._h)
        ns = H * D + H + L * (H * H + H) + D * H + D
        self.num_params = 2 * B * ns + (B if self.scale == 'constant' else 0)
        if Hn == H:
            assert self.num_params == native
            return self._set_pad_index(None, native)
        nn_ = Hn * D + Hn + L * (Hn * Hn + Hn) + D * Hn + D
        blocks = []
        for k in range(2 * B):
            off = k * nn_
            for su, sn in [((H, D), (Hn, D)), ((H,), (Hn,))] + [((H, H), (Hn, Hn)), ((H,), (Hn,))] * L + [((D, H), (D, Hn)), ((D,), (D,))]:
                blocks.append((su, sn, off))
                off += int(np.prod(sn))
        if self.scale == 'constant':
            blocks += [((), (), 2 * B * nn_ + b) for b in range(B)]
        self._set_pad_index(pad_index(blocks), native)
        assert len(self._pidx) == self.num_params

    # ---- weights ---------------------------------------------------------------------------------
    def layer_shapes(self):
        """[(name, shape, offset into the packed vector)] in torch state_dict order; with
        scale='translate'/'constant' a block has no scale_net, and with 'constant' the flows alternate coupling /
        ScaleLayer (networks.py:336-346)"""
        out = []
        D, H, L = self.D, self.H, self.L
        ns = H * D + H + L * (H * H + H) + D * H + D
        stride = 2 if self.scale == 'constant' else 1
        for b in range(self.B):
            for n, net in enumerate(('scale_net', 'translate_net')):
                if n == 0 and self.scale != '':
                    continue
                off = (2 * b + n) * ns
                dims = [(H, D)] + [(H, H)] * L + [(D, H)]
                for i, (o, k) in enumerate(dims):
                    out.append(('flow.flows.%d.%s.%d.weight' % (stride * b, net, 2 * i), (o, k), off))
                    off += o * k
                    out.append(('flow.flows.%d.%s.%d.bias' % (stride * b, net, 2 * i), (o,), off))
                    off += o
            if self.scale == 'constant':
                out.append(('flow.flows.%d.scale' % (2 * b + 1), (), 2 * self.B * ns + b))
        return out

    def default_init(self, seed=None):
        """nn.Linear's default init (kaiming_uniform(a=sqrt(5)) = U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for both
        weight and bias), which is what the reference ends up with: its orthogonal init closure is never
        applied (networks.py:284-287)."""
        g = torch.Generator()
        if seed is None:
            g.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
        else:
            g.manual_seed(int(seed))
        packed = np.zeros(self.num_params, np.float32)
        for name, shape, off in self.layer_shapes():
            if len(shape) == 0:
                continue                     # ScaleLayer scalar starts at 0 (networks.py:316)
            fan_in = shape[1] if len(shape) == 2 else None
            if fan_in is None:
                fan_in = last_fan_in
            else:
                last_fan_in = fan_in
            bound = 1.0 / math.sqrt(fan_in)
            n = int(np.prod(shape))
            packed[off:off + n] = ((torch.rand(n, generator=g) * 2 - 1) * bound).numpy()
        return packed

    def load_packed(self, packed):
        packed = np.ascontiguousarray(packed, dtype=np.float32)
        if packed.size != self.num_params:
            raise ValueError('expected %d packed weights, got %d' % (self.num_params, packed.size))
        packed = self._to_native(packed)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.nnest_nvp_load_weights(self._h, packed.ctypes.data_as(ctypes.c_void_p),
                                                        _lib.current_stream(self.device)))

    def store_packed(self):
        out = self._native_buffer()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.nnest_nvp_store_weights(self._h, out.ctypes.data_as(ctypes.c_void_p),
                                                         _lib.current_stream(self.device)))
        return self._from_native(out)

    def state_dict(self):
        return self.state_dict_from_packed(self.store_packed())

    def state_dict_from_packed(self, packed):
        """the reference's state_dict out of a packed-layout vector (no device access: safe on a worker thread)"""
        sd = {}
        for name, shape, off in self.layer_shapes():
            n = int(np.prod(shape))
            sd[name] = torch.from_numpy(packed[off:off + n].reshape(shape).copy())
        return sd

    def load_state_dict(self, sd):
        packed = np.zeros(self.num_params, np.float32)
        for name, shape, off in self.layer_shapes():
            v = sd[name]
            v = np.asarray(v.detach().cpu().numpy() if torch.is_tensor(v) else v, dtype=np.float32).ravel()
            packed[off:off + v.size] = v
        self.load_packed(packed)

    def reference_vector(self, packed=None):
        """the concatenated state_dict (what the reference's parameters() hold) out of a packed-layout vector"""
        packed = self.store_packed() if packed is None else np.asarray(packed)
        return np.concatenate([packed[off:off + int(np.prod(shape))] for _, shape, off in self.layer_shapes()])

    def load_reference_vector(self, vec):
        packed = np.zeros(self.num_params, np.float32)
        pos = 0
        for _, shape, off in self.layer_shapes():
            n = int(np.prod(shape))
            packed[off:off + n] = vec[pos:pos + n]
            pos += n
        self.load_packed(packed)

    def adam_step_count(self):
        n = ctypes.c_int(0)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.nnest_nvp_adam_state(self._h, ctypes.byref(n), -1, 0, _lib.current_stream(self.device)))
        return n.value

    def adam_moments(self):
        """host copies of Adam's (exp_avg, exp_avg_sq)"""
        m = self._native_buffer()
        v = self._native_buffer()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.nnest_nvp_store_adam(self._h, m.ctypes.data_as(ctypes.c_void_p),
                                                      v.ctypes.data_as(ctypes.c_void_p), _lib.current_stream(self.device)))
        return self._from_native(m), self._from_native(v)

    def set_adam(self, m, v, step):
        m = self._to_native(m)
        v = self._to_native(v)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.nnest_nvp_load_adam(self._h, m.ctypes.data_as(ctypes.c_void_p),
                                                     v.ctypes.data_as(ctypes.c_void_p), _lib.current_stream(self.device)))
            _lib.check(self._lib.nnest_nvp_adam_state(self._h, None, int(step), 0, _lib.current_stream(self.device)))

    def loss_grad(self, x):
        """loss = -mean(log_probs(x)) and dloss/dw (packed order) for one minibatch of <= 128 rows
        (loss.backward(), trainer.py:394-400), no weight update."""
        x = _as_dev_f32(x, self.device)
        grad = torch.empty(self._native_params, dtype=torch.float32, device=self.device)
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.nnest_nvp_loss_grad(self._h, _lib.ptr(x), x.shape[0], _lib.ptr(grad), _lib.ptr(loss),
                                                     _lib.current_stream(self.device)))
        return loss, self._from_native_dev(grad)

    def vjp(self, x, gz, gld):
        """the flow as one stage of a composite model (nnest_nvp_vjp): upstream gradient gz [M,D] and dL/d(logdet) in,
        dL/dw (packed order) and dL/dx out"""
        M = x.shape[0]
        grad = torch.empty(self._native_params, dtype=torch.float32, device=self.device)
        gx = torch.empty_like(x)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.nnest_nvp_vjp(self._h, _lib.ptr(x), _lib.ptr(gz.contiguous()), ctypes.c_float(gld), M, _lib.ptr(grad),
                                               _lib.ptr(gx), _lib.current_stream(self.device)))
        return self._from_native_dev(grad), gx

    def adam_step(self, grad, lr, weight_decay):
        """one torch.optim.Adam step (coupled weight decay, trainer.py:121-122) from a gradient in packed order"""
        grad = self._to_native_dev(grad)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.nnest_nvp_adam_step(self._h, _lib.ptr(grad), ctypes.c_float(lr), ctypes.c_float(weight_decay),
                                                     _lib.current_stream(self.device)))

    def train_epochs(self, xtrain, xvalid, perm, noise=None, seed=0, jitter=0.0, batch=100, max_epochs=1, patience=50,
                     lr=1e-3, weight_decay=1e-6, epoch_offset=0, resume=False, finalize=True, result=None, one_cu=False):
        """K5: Trainer.train's epoch loop (trainer.py:198-241) in one launch.  perm int32 [max_epochs, n_train];
        noise None (in-kernel Philox) or float32 [max_epochs, n_train, D] in loader order.
        A long train() can be split into chunks: pass the previous chunk's `result` tensor with resume=True,
        epoch_offset = epochs already run, finalize=True only for the last chunk (see include/nnest_hip.h).
        Returns dict(losses [epochs,2] tensor, epochs_run, best_epoch, best_validation_loss, last_train_loss,
        counter, stopped, result)."""
        dev = self.device
        if int(batch) > TRAIN_KERNEL_MAX_BATCH:   # the reference takes any batch_size (trainer.py:36, :76, :185): a slower path, not a refusal
            assert not resume and epoch_offset == 0, 'batch_size > %d: the host-driven loop takes a run in one call' % TRAIN_KERNEL_MAX_BATCH
            return train_epochs_host(self, xtrain, xvalid, perm, noise, seed, jitter, batch, max_epochs, patience, lr, weight_decay,
                                     chunk_rows=TRAIN_KERNEL_MAX_BATCH)
        xtrain = _as_dev_f32(xtrain, dev)
        xvalid = _as_dev_f32(xvalid, dev)
        perm = perm.to(device=dev, dtype=torch.int32).contiguous()
        n_train, n_valid = xtrain.shape[0], xvalid.shape[0]
        assert perm.numel() == max_epochs * n_train
        if noise is not None:
            noise = noise.to(device=dev, dtype=torch.float32).contiguous()
            assert noise.numel() == max_epochs * n_train * self.D
        losses = torch.zeros(max(max_epochs, 1), 2, dtype=torch.float32, device=dev)
        if result is None:
            assert not resume
            result = torch.zeros(6, dtype=torch.int32, device=dev)  # nnest_train_result_t
        flags = (_lib.TRAIN_RESUME if resume else 0) | (_lib.TRAIN_FINALIZE if finalize else 0) | (_lib.TRAIN_ONE_CU if one_cu else 0)
        with torch.cuda.device(dev):
            _lib.check(self._lib.nnest_nvp_train(self._h, _lib.ptr(xtrain), n_train, _lib.ptr(xvalid), n_valid,
                                                 _lib.ptr(perm), _lib.ptr(noise), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                                 float(jitter), int(batch), int(max_epochs), int(patience), float(lr),
                                                 float(weight_decay), int(epoch_offset), flags, _lib.ptr(losses),
                                                 _lib.ptr(result), _lib.current_stream(dev)))
        r = result.cpu()
        if int(r[5]) == 2:
            raise _lib.NnestHipError('nnest_nvp_train: a grid barrier of the multi-CU training kernel ran out (results invalid; error word 0x%x)' % (int(r[4]) & 0xffffffff))
        fl = r[2:4].view(torch.float32)
        return dict(losses=losses, epochs_run=int(r[0]), best_epoch=int(r[1]), best_validation_loss=float(fl[0]),
                    last_train_loss=float(fl[1]), counter=int(r[4]), stopped=bool(int(r[5])), result=result)


def loglike(like_id, x_unit, like_scale, device=None, like_params=None):
    """K6: batched analytic likelihood through safe_loglike (sampler.py:110-133): float64 [N]."""
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    x = _as_dev_f32(x_unit, device)
    out = torch.empty(x.shape[0], dtype=torch.float64, device=device)
    with torch.cuda.device(device):
        lk = _lib.like_spec(like_id, like_scale, like_params)
        _lib.check(_lib.load().nnest_loglike(ctypes.byref(lk), _lib.ptr(x), _lib.ptr(out), x.shape[0],
                                             x.shape[1], _lib.current_stream(device)))
    return out
