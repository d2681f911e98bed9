"""ORACLE -- TEST INFRASTRUCTURE ONLY (numpy/ctypes face of oracle/nnest_oracle.c).

Importable by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only; the product
package nnest_amd/ never imports it.  Parity status: PINNED (tests/test_oracle_golden.py checks
every function below against fixtures produced by running the reference, tests/golden/).
Each method names the reference function it restates (paths under /root/reference).
"""
import os
import ctypes
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libnnest_oracle.so')

LIKE_IDS = {'rosenbrock': 0, 'gaussmix': 1, 'mixture': 1, 'gaussianmix': 1, 'himmelblau': 2, 'gaussian': 3,
            'eggbox': 4, 'shell': 5, 'double_shell': 6}


def build(force=False):
    """Compile the C restatement (gcc; seconds)."""
    src = [os.path.join(_HERE, f) for f in ('nnest_oracle.c', 'nnest_oracle_impl.h', 'spline_oracle_impl.h', 'Makefile')]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(['make', '-C', _HERE, '--no-print-directory'], stdout=subprocess.DEVNULL)
    return _SO


_lib = None
_fp = ctypes.POINTER(ctypes.c_float)
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.orc_num_params.restype = ctypes.c_int
        for name in ('orc_loglike_f32', 'orc_loglike_f64', 'orc_train_step', 'orc_loss_grad_f64',
                     'orc_valid_loss', 'orc_training_jitter', 'orc32_nvp_loss_grad', 'orc64_nvp_loss_grad'):
            getattr(L, name).restype = ctypes.c_double
        L.orc_mcmc_sample.restype = ctypes.c_int
        L.orc_spline_num_params.restype = ctypes.c_int
        L.spl32_spline_log_probs.restype = ctypes.c_double
        L.spl64_spline_log_probs.restype = ctypes.c_double
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t):
    return a.ctypes.data_as(t)


SCALE_MODES = {'': 0, 'translate': 1, 'constant': 2}


def num_params(D, H, B, L, scale=''):
    lib().orc_set_scale_mode(SCALE_MODES[scale])
    return lib().orc_num_params(D, H, B, L)


def reference_index_map(D, H, B, L, scale=''):
    """index into the packed vector of every entry of the reference's state_dict, in state_dict order
    (networks.py:328-347: with scale='translate'/'constant' a block has no scale_net; with 'constant' a ScaleLayer
    scalar follows each coupling block)"""
    ns = H * D + H + L * (H * H + H) + D * H + D
    if scale == '':
        return np.arange(B * 2 * ns)
    idx = []
    for b in range(B):
        idx.append(np.arange(ns) + (2 * b + 1) * ns)
        if scale == 'constant':
            idx.append(np.array([B * 2 * ns + b]))
    return np.concatenate(idx)


class NVP(object):
    """RealNVP coupling stack with packed (state_dict-order) fp32 weights.

    Restates SingleSpeedNVP / NormalizingFlowModel (nnest/networks.py:17-84, :248-347)."""

    def __init__(self, D, H=16, B=3, L=1, weights=None, scale='', base_beta=0.0, kind='nvp'):
        self.D, self.H, self.B, self.L = int(D), int(H), int(B), int(L)
        self.scale = scale
        assert kind in ('nvp', 'maf') and (kind == 'nvp' or scale == '')
        self.kind = kind   # 'maf': the build-defined masked autoregressive flow (maf_oracle_impl.h; UNPINNED), same parameter layout
        self.base_beta = float(base_beta)   # 0: N(0,I); > 0: GeneralisedNormal(0, 1, beta) (distributions/generalised_normal.py)
        self.n = num_params(D, H, B, L, scale)
        self.w = np.zeros(self.n, np.float32) if weights is None else _f32(weights).copy()
        assert self.w.size == self.n, (self.w.size, self.n)
        self.m = np.zeros(self.n, np.float32)  # Adam exp_avg
        self.v = np.zeros(self.n, np.float32)  # Adam exp_avg_sq
        self.t = 0                             # Adam step count

    def _cfg(self):
        lib().orc_set_scale_mode(SCALE_MODES[self.scale])
        lib().orc_set_flow_kind(1 if self.kind == 'maf' else 0)
        lib().orc_set_base_beta(ctypes.c_double(self.base_beta))
        return (_p(self.w, _fp), self.D, self.H, self.B, self.L)

    def load_reference_vector(self, vec):
        """weights given as the concatenated reference state_dict (scale variants: no scale nets)"""
        self.w[:] = 0
        self.w[reference_index_map(self.D, self.H, self.B, self.L, self.scale)] = _f32(vec)

    def reference_vector(self, arr=None):
        return (self.w if arr is None else arr)[reference_index_map(self.D, self.H, self.B, self.L, self.scale)]

    def _run(self, fn32, fn64, x, f64, two_out=True):
        x = np.atleast_2d(x)
        N = x.shape[0]
        if f64:
            xi = _f64(x); o = np.empty((N, self.D)); ld = np.empty(N); t = _dp
            fn = fn64
        else:
            xi = _f32(x); o = np.empty((N, self.D), np.float32); ld = np.empty(N, np.float32); t = _fp
            fn = fn32
        fn(*self._cfg(), _p(xi, t), N, _p(o, t), _p(ld, t))
        return o, ld

    def forward(self, x, f64=False):
        """NormalizingFlow.forward (networks.py:24-32) -> (z, logdet)"""
        return self._run(lib().orc32_nvp_forward, lib().orc64_nvp_forward, x, f64)

    def inverse(self, z, f64=False):
        """NormalizingFlow.inverse (networks.py:34-42) -> (x, logdet)"""
        return self._run(lib().orc32_nvp_inverse, lib().orc64_nvp_inverse, z, f64)

    def log_probs(self, x, f64=False):
        """NormalizingFlowModel.log_probs (networks.py:71-76)"""
        x = np.atleast_2d(x)
        N = x.shape[0]
        if f64:
            xi = _f64(x); o = np.empty(N)
            lib().orc64_nvp_log_probs(*self._cfg(), _p(xi, _dp), N, _p(o, _dp))
        else:
            xi = _f32(x); o = np.empty(N, np.float32)
            lib().orc32_nvp_log_probs(*self._cfg(), _p(xi, _fp), N, _p(o, _fp))
        return o

    def loss_grad(self, X, f64=False):
        """loss = -mean(log_probs(X)) and its gradient wrt the packed weights (trainer.py:394, :400)"""
        X = _f32(X)
        M = X.shape[0]
        if f64:
            g = np.empty(self.n)
            loss = lib().orc_loss_grad_f64(*self._cfg(), _p(X, _fp), M, _p(g, _dp))
        else:
            g = np.empty(self.n, np.float32)
            loss = lib().orc32_nvp_loss_grad(*self._cfg(), _p(X, _fp), M, _p(g, _fp))
        return loss, g

    def train_step(self, X, idx, noise, jitter, lr=1e-3, wd=1e-6):
        """One minibatch of Trainer._train (trainer.py:390-401) + Adam (trainer.py:121-122)."""
        X = _f32(X)
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        M = idx.size
        nz = None if noise is None else _f32(noise)
        g = np.empty(self.n, np.float32)
        self.t += 1
        self._cfg()
        loss = lib().orc_train_step(_p(self.w, _fp), _p(self.m, _fp), _p(self.v, _fp), self.D, self.H, self.B,
                                    self.L, _p(X, _fp), _p(idx, _ip), None if nz is None else _p(nz, _fp), M,
                                    ctypes.c_float(jitter), self.t, ctypes.c_float(lr), ctypes.c_float(wd),
                                    _p(g, _fp))
        return loss, g

    def valid_loss(self, X):
        """Trainer._validate's batch loss before the /len(dataset) (trainer.py:405-418)"""
        X = _f32(X)
        return lib().orc_valid_loss(*self._cfg(), _p(X, _fp), X.shape[0])

    def train_epoch(self, X, perm, noise, jitter, batch=100, lr=1e-3, wd=1e-6):
        """Trainer._train (trainer.py:384-403): returns sum(batch means)/len(dataset) like the reference."""
        n = X.shape[0]
        tot = 0.0
        for b in range(0, n, batch):
            idx = perm[b:b + batch]
            nz = None if noise is None else noise[b:b + batch]
            loss, _ = self.train_step(X, idx, nz, jitter, lr, wd)
            tot += loss
        return tot / n

    def train(self, samples, perm_split, perms, noises, jitter, max_iters, patience=50, batch=100, lr=1e-3,
              wd=1e-6, validation_fraction=0.1):
        """Trainer.train (trainer.py:134-245) with the split / shuffles / jitter noise supplied."""
        N = samples.shape[0]
        n_valid = int(np.ceil(validation_fraction * N))
        n_train = N - n_valid
        Xv = _f32(samples[perm_split[:n_valid]])
        Xt = _f32(samples[perm_split[n_valid:n_valid + n_train]])
        best, best_epoch, best_w, counter = float('inf'), 0, self.w.copy(), 0
        tl, vl = [], []
        for epoch in range(1, max_iters + 1):
            tl.append(self.train_epoch(Xt, perms[epoch - 1], None if noises is None else noises[epoch - 1],
                                       jitter, batch, lr, wd))
            v = self.valid_loss(Xv) / n_valid
            vl.append(v)
            if v < best:
                best, best_epoch, best_w, counter = v, epoch, self.w.copy(), 0
            counter += 1
            if counter > patience:
                break
        self.w = best_w
        return dict(train_losses=np.array(tl), valid_losses=np.array(vl), best_validation_loss=best,
                    best_validation_epoch=best_epoch, epochs_run=len(tl))


class Spline(object):
    """SingleSpeedSpline (nnest/networks.py:708-715): [ActNorm, Invertible1x1Conv, NSF_CL] x B with packed
    (state_dict-order) fp32 weights and the B fixed permutation matrices P (not part of the state_dict)."""

    def __init__(self, D, H=16, B=3, K=8, tail=3.0, weights=None, P=None, base_beta=0.0):
        self.D, self.H, self.B, self.K, self.tail = int(D), int(H), int(B), int(K), float(tail)
        self.base_beta = float(base_beta)
        self.n = lib().orc_spline_num_params(self.D, self.H, self.B, self.K)
        self.w = np.zeros(self.n, np.float32) if weights is None else _f32(weights).copy()
        assert self.w.size == self.n, (self.w.size, self.n)
        self.P = _f32(np.tile(np.eye(self.D), (self.B, 1, 1)) if P is None else P).reshape(self.B, self.D, self.D).copy()

    def _args(self, f64):
        lib().orc_set_base_beta(ctypes.c_double(self.base_beta))
        t = ctypes.c_double(self.tail) if f64 else ctypes.c_float(self.tail)
        return (_p(self.w, _fp), _p(self.P, _fp), self.D, self.H, self.B, self.K, t)

    def forward(self, x, f64=False, data_init=False):
        """NormalizingFlow.forward (networks.py:24-32); data_init: ActNorm's first-batch initialisation
        (networks.py:698-705), written into self.w"""
        x = np.atleast_2d(x)
        N = x.shape[0]
        if f64:
            xi = _f64(x); z = np.empty((N, self.D)); ld = np.empty(N)
            lib().spl64_spline_forward(*self._args(True), _p(xi, _dp), N, _p(z, _dp), _p(ld, _dp), int(data_init))
        else:
            xi = _f32(x); z = np.empty((N, self.D), np.float32); ld = np.empty(N, np.float32)
            lib().spl32_spline_forward(*self._args(False), _p(xi, _fp), N, _p(z, _fp), _p(ld, _fp), int(data_init))
        return z, ld

    def inverse(self, z, f64=False):
        """NormalizingFlow.inverse (networks.py:34-42)"""
        z = np.atleast_2d(z)
        N = z.shape[0]
        if f64:
            zi = _f64(z); x = np.empty((N, self.D)); ld = np.empty(N)
            lib().spl64_spline_inverse(*self._args(True), _p(zi, _dp), N, _p(x, _dp), _p(ld, _dp))
        else:
            zi = _f32(z); x = np.empty((N, self.D), np.float32); ld = np.empty(N, np.float32)
            lib().spl32_spline_inverse(*self._args(False), _p(zi, _fp), N, _p(x, _fp), _p(ld, _fp))
        return x, ld

    def log_probs(self, x, f64=False, data_init=False):
        """NormalizingFlowModel.log_probs (networks.py:71-76) -> (log_probs, -mean)"""
        x = np.atleast_2d(x)
        N = x.shape[0]
        if f64:
            xi = _f64(x); lp = np.empty(N)
            loss = lib().spl64_spline_log_probs(*self._args(True), _p(xi, _dp), N, _p(lp, _dp), int(data_init))
        else:
            xi = _f32(x); lp = np.empty(N, np.float32)
            loss = lib().spl32_spline_log_probs(*self._args(False), _p(xi, _fp), N, _p(lp, _fp), int(data_init))
        return lp, loss

    def fd_grad(self, X, idx, h=5e-5):
        """central finite differences (float64 arithmetic, float32 weights) of loss = -mean(log_probs(X)) for the
        packed parameters `idx`: the yardstick for hand-written backward passes"""
        out = np.empty(len(idx))
        for k, i in enumerate(idx):
            w0 = self.w[i]
            self.w[i] = np.float32(w0 + h); hp = float(self.w[i]) - float(w0)
            lp_ = self.log_probs(X, f64=True)[1]
            self.w[i] = np.float32(w0 - h); hm = float(w0) - float(self.w[i])
            lm_ = self.log_probs(X, f64=True)[1]
            self.w[i] = w0
            out[k] = (lp_ - lm_) / (hp + hm)
        return out


def loglike(name, x_unit, scale, params=None):
    """safe_loglike(x) = loglike(transform(x)) (nnest/sampler.py:110-133; nnest/likelihoods.py).
    float32 input follows the reference's float32 arithmetic, float64 input its float64 arithmetic."""
    lid = LIKE_IDS[name.lower()]
    x = np.atleast_2d(x_unit)
    N, D = x.shape
    out = np.empty(N)
    if lid >= 3:
        par = np.zeros(6)
        par[:len(params or ())] = params or ()
        xi = _f64(x)
        lib().orc_loglike2_batch(lid, _p(xi, _dp), N, D, ctypes.c_double(scale), _p(par, _dp),
                                 int(x.dtype == np.float32), _p(out, _dp))
        return out
    if x.dtype == np.float32:
        xi = _f32(x)
        lib().orc_loglike_batch_f32(lid, _p(xi, _fp), N, D, ctypes.c_float(scale), _p(out, _dp))
    else:
        xi = _f64(x)
        lib().orc_loglike_batch_f64(lid, _p(xi, _dp), N, D, ctypes.c_double(scale), _p(out, _dp))
    return out


def prior_inbox(x):
    """UniformPrior(D,-1,1) via safe_prior (nnest/priors.py:39-43, nnest/sampler.py:152-161): 0 / -inf"""
    x = np.atleast_2d(x)
    if x.dtype == np.float32:
        xi = _f32(x)
        flag = [lib().orc_prior_inbox_f32(_p(xi[i], _fp), x.shape[1]) for i in range(x.shape[0])]
    else:
        xi = _f64(x)
        flag = [lib().orc_prior_inbox_f64(_p(xi[i], _dp), x.shape[1]) for i in range(x.shape[0])]
    return np.where(np.array(flag) == 1, 0.0, -np.inf)


def mcmc_sample(nvp, like, like_scale, init, init_logl, loglstar, step, dynamic, dz, u, lag=0, margins=None, warm=0):
    """Sampler._mcmc_sample hard-constraint branch (nnest/sampler.py:229-463) with recorded noise.
    Returns the reference's tuple pieces: samples, latent, loglikes, scale, ncall, (acc, rej).
    lag > 0: build-defined variant of the step-size rule -- the update after step `it` uses the accepted count of step
    it - lag (what the GPU's batch-wide mode does to keep the grid-wide wait off the step; 0 = the reference).
    margins: optional float64 [S, C] array that receives, per step and walker, how far the step's decision was from its
    thresholds (orc_set_margin_out, nnest_oracle.c) -- for asserting that a kernel's differing decisions were borderline."""
    S, C, D = dz.shape
    init = _f64(init); init_logl = _f64(init_logl); dz = _f32(dz); u = _f32(u)
    samples = np.empty((C, S + 1, D), np.float32)
    latent = np.empty((C, S + 1, D), np.float32)
    loglikes = np.empty((C, S + 1))
    scale = ctypes.c_double(step)
    acc = ctypes.c_long(0); rej = ctypes.c_long(0)
    lib().orc_set_scale_mode(SCALE_MODES[nvp.scale])
    lib().orc_set_flow_kind(1 if getattr(nvp, 'kind', 'nvp') == 'maf' else 0)
    lib().orc_set_step_lag(int(lag))
    lib().orc_set_step_warm(int(warm))   # first `warm` steps exact in front of the lagged ones (NNEST_MH_WARM)
    if margins is not None:
        assert margins.shape == (S, C) and margins.dtype == np.float64 and margins.flags['C_CONTIGUOUS']
        lib().orc_set_margin_out(_p(margins, _dp))
    ncall = lib().orc_mcmc_sample(_p(nvp.w, _fp), nvp.D, nvp.H, nvp.B, nvp.L, LIKE_IDS[like.lower()],
                                  ctypes.c_float(like_scale), _p(init, _dp), _p(init_logl, _dp), C, S,
                                  ctypes.c_double(loglstar), ctypes.byref(scale), int(bool(dynamic)), _p(dz, _fp),
                                  _p(u, _fp), _p(samples, _fp), _p(latent, _fp), _p(loglikes, _dp),
                                  ctypes.byref(acc), ctypes.byref(rej))
    lib().orc_set_step_lag(0)
    lib().orc_set_step_warm(0)
    lib().orc_set_margin_out(None)
    return samples, latent, loglikes, scale.value, ncall, (acc.value, rej.value)


def training_jitter(samples):
    """trainer.py:168-171 with jitter < 0: 0.2 * mean(cKDTree(samples).query(samples, 2) distances)"""
    X = _f64(samples)
    return lib().orc_training_jitter(_p(X, _dp), X.shape[0], X.shape[1])


def philox4x32_10(ctr, key):
    c = (ctypes.c_uint32 * 4)(*[int(v) & 0xffffffff for v in ctr])
    k = (ctypes.c_uint32 * 2)(*[int(v) & 0xffffffff for v in key])
    o = (ctypes.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(v) for v in o]


def slice_uniform(seed, walker, idx):
    """the slice kernel's uniform draw k of step s (idx = 64 s + k): Philox4x32-10 of (seed; walker, idx), stream U, 24 bits
    (flow_tile.h noise_uniform) -- exact in float32"""
    seed, walker = int(seed) & 0xFFFFFFFFFFFFFFFF, int(walker) & 0xFFFFFFFFFFFFFFFF
    ctr = [0, walker & 0xffffffff, int(idx) & 0xffffffff, ((walker >> 32) & 0x0fffffff) | (1 << 28)]
    r = philox4x32_10(ctr, [seed & 0xffffffff, seed >> 32])
    return np.float32((r[0] >> 8) * 5.9604644775390625e-08)


def slice_sample(flow, like, like_scale, z0, logl0, loglstar, width, dz, seed, walker_offset=0, max_stepout=8, max_shrink=32,
                 margins=None):
    """[BUILD-DEFINED, parity unpinned: the reference proposes random-walk Metropolis moves only, nnest/sampler.py:310-316]
    CPU restatement of the slice proposal kernel (nnest_amd/csrc/nnest_solo.hip slice_kernel_solo; include/nnest_hip.h
    nnest_slice_steps): univariate slice sampling (stepping out + shrinkage) along the recorded directions dz [S, C, D] of the target
    |det dx/dz| on {x(z) in the unit box, logL > loglstar}; `flow` is an oracle flow object (NVP).  Returns the per-step x history
    [C, S + 1, D], the final z, logl, and the counters.  margins [S, C] (optional): the smallest distance of any evaluated candidate of
    the step from a decision threshold (box edge, slice level, L*) -- a chain that differs from the kernel's must have come close."""
    S, C, D = dz.shape
    z = np.asarray(z0, dtype=np.float32).copy()
    x, ld = flow.inverse(z)
    logl = np.asarray(logl0, dtype=np.float64).copy()
    hx = np.empty((C, S + 1, D), np.float32)
    hx[:, 0] = x
    n_call, n_move, n_eval = np.zeros(C, int), np.zeros(C, int), np.zeros(C, int)
    w32 = np.float32(width)
    for c in range(C):
        zc, xc, ldc, lc = z[c].copy(), x[c].copy(), np.float32(ld[c]), float(logl[c])
        for it in range(1, S + 1):
            e = dz[it - 1, c].astype(np.float32)
            u0, u1 = slice_uniform(seed, walker_offset + c, 64 * it), slice_uniform(seed, walker_offset + c, 64 * it + 1)
            with np.errstate(divide='ignore'):
                logy = np.float32(ldc + np.log(u1, dtype=np.float32))
            mg = [np.inf]

            def inside(t):
                tw = np.float32(np.float32(t) * w32)
                zp = (zc.astype(np.float64) + e.astype(np.float64) * np.float64(tw)).astype(np.float32)   # one fused multiply-add per dim
                xp, ldp = flow.inverse(zp[None])
                xp, ldp = xp[0], np.float32(ldp[0])
                inb = bool(prior_inbox(xp[None])[0] == 0)
                pre = inb and bool(ldp > logy)
                lp = float(loglike(like, xp[None], like_scale)[0])
                n_eval[c] += 1
                n_call[c] += 1 if pre else 0
                mg[0] = min(mg[0], float(np.min(np.abs(np.abs(xp.astype(np.float64)) - 1.0))), abs(float(ldp) - float(logy)),
                            abs(lp - loglstar) / (1.0 + abs(loglstar)))
                return (pre and lp > loglstar), zp, xp, ldp, lp

            tl, tr = np.float32(-u0), np.float32(np.float32(1.0) - u0)
            for _ in range(max_stepout):
                if not inside(tl)[0]:
                    break
                tl = np.float32(tl - np.float32(1.0))
            for _ in range(max_stepout):
                if not inside(tr)[0]:
                    break
                tr = np.float32(tr + np.float32(1.0))
            for k in range(max_shrink):
                uk = slice_uniform(seed, walker_offset + c, 64 * it + 2 + k)
                t = np.float32(np.float64(np.float32(tr - tl)) * np.float64(uk) + np.float64(tl))   # fmaf(tr - tl, uk, tl)
                ok, zp, xp, ldp, lp = inside(t)
                if ok:
                    zc, xc, ldc, lc = zp, xp, ldp, lp
                    n_move[c] += 1
                    break
                if t < 0:
                    tl = t
                else:
                    tr = t
            hx[c, it] = xc
            if margins is not None:
                margins[it - 1, c] = mg[0]
        z[c], logl[c] = zc, lc
    return dict(x=hx, z=z, logl=logl, n_call=n_call, n_move=n_move, n_eval=n_eval)


class FastSlowNVP(object):
    """FastSlowNVP (nnest/networks.py:86-150, :350-380): slow NVP on x[:, :S], fast NVP on x[:, S:], then one coupling layer
    (hidden 64, one hidden layer) with mask = (1,)*S + (0,)*F.  Weights: the concatenated reference state_dict
    (fast_flow, slow_flow, flow).  Composition of the pinned NVP pieces; the block-mask coupling is restated in numpy."""

    def __init__(self, S, F, H=16, B=3, L=1, weights=None):
        self.S, self.F, self.D = int(S), int(F), int(S) + int(F)
        self.fast = NVP(F, H, B, L)
        self.slow = NVP(S, H, B, L)
        D, Hc = self.D, 64
        self.cshapes = [(Hc, D), (Hc,), (Hc, Hc), (Hc,), (D, Hc), (D,)]
        self.n = self.fast.n + self.slow.n + 2 * sum(int(np.prod(s)) for s in self.cshapes)
        if weights is not None:
            self.load(weights)

    def load(self, w):
        w = _f32(w)
        assert w.size == self.n
        self.fast.w[:] = w[:self.fast.n]
        self.slow.w[:] = w[self.fast.n:self.fast.n + self.slow.n]
        off = self.fast.n + self.slow.n
        self.c = {}
        for net in ('scale', 'translate'):
            self.c[net] = []
            for shp in self.cshapes:
                k = int(np.prod(shp))
                self.c[net].append(w[off:off + k].reshape(shp))
                off += k

    def _mlp(self, net, m, act, dt):
        W0, b0, W1, b1, W2, b2 = [a.astype(dt) for a in self.c[net]]
        f = np.tanh if act == 'tanh' else (lambda v: np.maximum(v, 0))
        h = f(m @ W0.T + b0)
        h = f(h @ W1.T + b1)
        return h @ W2.T + b2

    def _coupling(self, y, inverse, dt):
        """CouplingLayer.forward / inverse (networks.py:289-309) with the slow|fast block mask"""
        mask = np.concatenate([np.ones(self.S), np.zeros(self.F)]).astype(dt)
        m = y * mask
        ls = self._mlp('scale', m, 'tanh', dt) * (1 - mask)
        t = self._mlp('translate', m, 'relu', dt) * (1 - mask)
        if inverse:
            return (y - t) * np.exp(-ls), -ls.sum(-1)
        return y * np.exp(ls) + t, ls.sum(-1)

    def forward(self, x, f64=False):
        dt = np.float64 if f64 else np.float32
        x = np.atleast_2d(x).astype(dt)
        s, lds = self.slow.forward(x[:, :self.S], f64=f64)
        f, ldf = self.fast.forward(x[:, self.S:], f64=f64)
        z, ldc = self._coupling(np.concatenate([s, f], 1).astype(dt), False, dt)
        return z, lds + ldf + ldc

    def inverse(self, z, f64=False):
        dt = np.float64 if f64 else np.float32
        y, ldc = self._coupling(np.atleast_2d(z).astype(dt), True, dt)
        s, lds = self.slow.inverse(y[:, :self.S], f64=f64)
        f, ldf = self.fast.inverse(y[:, self.S:], f64=f64)
        return np.concatenate([s, f], 1), ldc + lds + ldf

    def log_probs(self, x, f64=False):
        z, ld = self.forward(x, f64=f64)
        return -0.5 * np.sum(z * z, axis=1) - 0.5 * self.D * np.log(2 * np.pi) + ld


class Cholesky(object):
    """SingleSpeedCholeksy (nnest/networks.py:162-239): y = L x + b, L lower triangular with diag = softplus(u) + 1e-3.
    Weights: bias[D], lower_entries[D(D-1)/2] (np.tril_indices(D, -1) order), unconstrained_diag[D].  numpy restatement."""

    def __init__(self, D, weights):
        self.D = int(D)
        self.w = _f32(weights).copy()

    def _parts(self, dt):
        D = self.D
        nl = D * (D - 1) // 2
        b, lo, ud = self.w[:D].astype(dt), self.w[D:D + nl].astype(dt), self.w[D + nl:].astype(dt)
        L = np.zeros((D, D), dt)
        L[np.tril_indices(D, -1)] = lo
        diag = np.log1p(np.exp(ud)) + dt(1e-3)
        L[np.diag_indices(D)] = diag
        return L, b, diag

    def forward(self, x, f64=False):
        dt = np.float64 if f64 else np.float32
        L, b, diag = self._parts(dt)
        x = np.atleast_2d(x).astype(dt)
        return x @ L.T + b, np.full(x.shape[0], np.sum(np.log(diag)), dt)

    def inverse(self, z, f64=False):
        dt = np.float64 if f64 else np.float32
        L, b, diag = self._parts(dt)
        z = np.atleast_2d(z).astype(dt)
        x = np.linalg.solve(L.astype(np.float64), (z - b).T.astype(np.float64)).T.astype(dt)
        return x, np.full(z.shape[0], -np.sum(np.log(diag)), dt)

    def log_probs(self, x, f64=False):
        y, ld = self.forward(x, f64=f64)
        return -0.5 * np.sum(y * y, axis=1) - 0.5 * self.D * np.log(2 * np.pi) + ld
