"""HipSpline: the neural-spline flow resident on one MI355X, driven through the C ABI.

Mirrors SingleSpeedSpline (reference nnest/networks.py:708-715): [ActNorm, Invertible1x1Conv, NSF_CL] x num_blocks,
rational-quadratic splines with num_bins = 8 on [-tail_bound, tail_bound].  forward / inverse / log_probs / sample and
the fused proposal kernel run as HIP kernels (nnest_amd/csrc/spline_tile.h); no PyTorch arithmetic is used for the flow.

state_dict() has the reference's keys and order (flow.flows.<3b>.s/.t, <3b+1>.L/.S/.U, <3b+2>.f1/f2.net.{0,2,4,6}).
The fixed permutations P of the 1x1 convolutions are plain attributes of the reference module (networks.py:634-635),
absent from its state_dict: they travel as `P` ([num_blocks, D, D]).
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib
from .flow import _HipFlow, _PaddedVectors, _as_dev_f32, pad_index, native_hidden, train_epochs_host, TRAIN_KERNEL_MAX_BATCH


class HipSpline(_PaddedVectors, _HipFlow):

    def __init__(self, num_inputs, hidden_dim=16, num_blocks=3, num_bins=8, tail_bound=3.0, device=None, seed=None):
        if not torch.cuda.is_available():
            raise _lib.NnestHipError('HipSpline needs an MI355X visible to PyTorch-ROCm (torch.cuda.is_available() is '
                                     'False); there is no CPU fallback')
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.D, self.H, self.B, self.K = int(num_inputs), int(hidden_dim), int(num_blocks), int(num_bins)
        self.tail_bound = float(tail_bound)
        self.num_inputs = self.D
        self.nu = self.D // 2
        self.nl = self.D - self.nu
        self._lib = _lib.load()
        L = self._lib
        self._sym = dict(forward=L.nnest_spline_forward, inverse=L.nnest_spline_inverse, log_probs=L.nnest_spline_log_probs,
                         inverse_loglike=L.nnest_spline_inverse_loglike, mh=L.nnest_spline_mh_constrained_steps,
                         set_base=L.nnest_spline_set_base)
        self._h = ctypes.c_void_p()
        self._Hn = native_hidden(self.H)     # the native handle's hidden width (flow._PaddedVectors: zero-padded, exact)
        with torch.cuda.device(self.device):
            _lib.check(L.nnest_spline_create(self.D, self._Hn, self.B, self.K, ctypes.c_float(self.tail_bound),
                                             ctypes.byref(self._h)))
        native = L.nnest_spline_num_params(self._h)
        self.num_params = sum(int(np.prod(shape)) for _, shape in self.layer_shapes())
        if self._Hn == self.H:
            assert self.num_params == native
            self._set_pad_index(None, native)
        else:
            H, Hn, blocks, off = self.H, self._Hn, [], 0
            for _, su, axes in self._layer_shapes_axes():
                sn = tuple(Hn if k in axes else d for k, d in enumerate(su))
                blocks.append((su, sn, off))
                off += int(np.prod(sn))
            assert off == native
            self._set_pad_index(pad_index(blocks), native)
        self.prior = torch.distributions.MultivariateNormal(torch.zeros(self.D, device=self.device),
                                                            torch.eye(self.D, device=self.device))
        self.data_dep_init_done = False
        w, P = self.default_init(seed)
        self.load_packed(w, P)

    SPLINE_MH_FORMS = {0: 'wave', 1: 'team', 2: 'pair'}

    def kernel_form_for(self, C, dynamic=False, lag=None):
        """the form of the proposal kernel `mh_steps` runs for C walkers under this step rule (nnest_spline_mh_form_for): 'pair'
        (8 walkers per workgroup, each in both halves of the matrix-core columns), 'team' (four waves per 16 walkers), 'wave' (one
        wave per 16 walkers); None if the launch would be refused.  (Not `mh_form_for`: the spline forms cannot be pinned -- they
        agree to rounding, not to the bit.)"""
        f = self._lib.nnest_spline_mh_form_for(self._h, int(C), _lib.mh_flags(dynamic, False, lag, None, 0))
        return self.SPLINE_MH_FORMS.get(f)

    def train_form_for(self, batch):
        """'rows' (one row of the minibatch per workgroup, nnest_spline_rows.hip) or 'tiles' (nnest_spline_train.hip): what a minibatch
        of `batch` rows runs in (nnest_spline_train_form)"""
        return 'rows' if self._lib.nnest_spline_train_form(self._h, int(batch)) == 1 else 'tiles'

    def __del__(self):
        try:
            if getattr(self, '_h', None) is not None and self._h.value:
                self._lib.nnest_spline_destroy(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass

    # ---- weights ---------------------------------------------------------------------------------
    def layer_shapes(self):
        """[(name, shape)] in torch state_dict order"""
        return [(n, sh) for n, sh, _ in self._layer_shapes_axes()]

    def _layer_shapes_axes(self):
        """[(name, shape, axes of the shape that are the conditioners' hidden width)]"""
        out = []
        D, H, Pn = self.D, self.H, 3 * self.K - 1
        for b in range(self.B):
            out += [('flow.flows.%d.s' % (3 * b), (1, D), ()), ('flow.flows.%d.t' % (3 * b), (1, D), ())]
            out += [('flow.flows.%d.L' % (3 * b + 1), (D, D), ()), ('flow.flows.%d.S' % (3 * b + 1), (D,), ()),
                    ('flow.flows.%d.U' % (3 * b + 1), (D, D), ())]
            for f, nin, nout in (('f1', self.nl, Pn * self.nu), ('f2', self.nu, Pn * self.nl)):
                dims = [((H, nin), (0,), (0,)), ((H, H), (0, 1), (0,)), ((H, H), (0, 1), (0,)), ((nout, H), (1,), ())]
                for i, (sh, wax, bax) in enumerate(dims):
                    out.append(('flow.flows.%d.%s.net.%d.weight' % (3 * b + 2, f, 2 * i), sh, wax))
                    out.append(('flow.flows.%d.%s.net.%d.bias' % (3 * b + 2, f, 2 * i), (sh[0],), bax))
        return out

    def default_init(self, seed=None):
        """The reference's construction (networks.py:630-638, :669-670, nn.Linear defaults): ActNorm s, t ~ N(0,1)
        (replaced by the data-dependent initialisation on the first forward batch); 1x1 conv from the LU
        decomposition of a random orthogonal matrix; Linear layers U(-1/sqrt(fan_in), 1/sqrt(fan_in))."""
        g = torch.Generator()
        g.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()) if seed is None else int(seed))
        D = self.D
        parts, Ps = [], []
        for name, shape in self.layer_shapes():
            leaf = name.split('.')[-1]
            if leaf in ('s', 't'):
                parts.append(torch.randn(D, generator=g))
            elif leaf == 'L':
                Q = torch.linalg.qr(torch.randn(D, D, generator=g, dtype=torch.float64))[0]
                P, L, U = torch.linalg.lu(Q)
                Ps.append(P.to(torch.float32))
                self._lu = (L, U)
                parts.append(L.to(torch.float32).reshape(-1))
            elif leaf == 'S':
                parts.append(torch.diagonal(self._lu[1]).to(torch.float32))
            elif leaf == 'U':
                parts.append(torch.triu(self._lu[1], diagonal=1).to(torch.float32).reshape(-1))
            else:
                fan_in = shape[1] if len(shape) == 2 else last_fan_in
                last_fan_in = fan_in
                n = int(np.prod(shape))
                parts.append((torch.rand(n, generator=g) * 2 - 1) / math.sqrt(fan_in))
        return torch.cat(parts).numpy().astype(np.float32), torch.stack(Ps).numpy()

    def load_packed(self, packed, P=None):
        packed = np.ascontiguousarray(packed, dtype=np.float32)
        if packed.size != self.num_params:
            raise ValueError('expected %d packed weights, got %d' % (self.num_params, packed.size))
        packed = self._to_native(packed)
        pp = None
        if P is not None:
            P = np.ascontiguousarray(P, dtype=np.float32)
            if P.shape != (self.B, self.D, self.D):
                raise ValueError('P: expected %s, got %s' % ((self.B, self.D, self.D), P.shape))
            pp = P.ctypes.data_as(ctypes.c_void_p)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.nnest_spline_load_weights(self._h, packed.ctypes.data_as(ctypes.c_void_p), pp,
                                                           _lib.current_stream(self.device)))

    def store_packed(self):
        out = self._native_buffer()
        _lib.check(self._lib.nnest_spline_store_weights(self._h, out.ctypes.data_as(ctypes.c_void_p), None, None))
        return self._from_native(out)

    @property
    def P(self):
        out = np.empty((self.B, self.D, self.D), np.float32)
        _lib.check(self._lib.nnest_spline_store_weights(self._h, None, out.ctypes.data_as(ctypes.c_void_p), None))
        return out

    def state_dict(self):
        packed = self.store_packed()
        sd, off = {}, 0
        for name, shape in self.layer_shapes():
            n = int(np.prod(shape))
            sd[name] = torch.from_numpy(packed[off:off + n].reshape(shape).copy())
            off += n
        return sd

    def load_state_dict(self, sd, P=None):
        if P is None:
            P = self.P   # keep the permutations already loaded
        self.load_packed(np.concatenate([np.asarray(sd[name].detach().cpu().numpy() if torch.is_tensor(sd[name])
                                                    else sd[name], dtype=np.float32).ravel()
                                         for name, _ in self.layer_shapes()]), P)

    # ---- ActNorm's data-dependent initialisation (networks.py:698-705) -------------------------------
    # The reference initialises s, t of every ActNorm from the first batch that is pushed FORWARD through a freshly
    # constructed model (forward / log_probs / the first training minibatch); `data_dep_init_done` is a plain attribute
    # there too, so a state_dict loaded into a new object is re-initialised by its first forward batch -- kept as is.
    def actnorm_init(self, x):
        x = _as_dev_f32(x, self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.nnest_spline_actnorm_init(self._h, _lib.ptr(x), x.shape[0], _lib.current_stream(self.device)))
        self.data_dep_init_done = True

    def forward(self, x):
        if not self.data_dep_init_done:
            self.actnorm_init(x)
        return super().forward(x)

    def log_probs(self, x):
        if not self.data_dep_init_done:
            self.actnorm_init(x)
        return super().log_probs(x)

    # ---- training ---------------------------------------------------------------------------------
    epoch_chunk = 1 << 30   # Trainer.train hands the whole run to one call (the epoch loop is host-driven here)

    def loss_grad(self, x):
        """loss = -mean(log_probs(x)) and dloss/dw (packed order), no weight update (trainer.py:394-400)"""
        x = _as_dev_f32(x, self.device)
        grad = torch.empty(self._native_params, dtype=torch.float32, device=self.device)
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.nnest_spline_loss_grad(self._h, _lib.ptr(x), x.shape[0], _lib.ptr(grad), _lib.ptr(loss),
                                                        _lib.current_stream(self.device)))
        return loss, self._from_native_dev(grad)

    def vjp(self, x, gz, gld):
        """the flow as one stage of a composite model (nnest_spline_vjp): upstream gradient gz [M,D] and dL/d(logdet) in,
        dL/dw (packed order) and dL/dx out"""
        M = x.shape[0]
        grad = torch.empty(self._native_params, dtype=torch.float32, device=self.device)
        gx = torch.empty_like(x)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.nnest_spline_vjp(self._h, _lib.ptr(x), _lib.ptr(gz.contiguous()), ctypes.c_float(gld), M, _lib.ptr(grad),
                                                  _lib.ptr(gx), _lib.current_stream(self.device)))
        return self._from_native_dev(grad), gx

    def adam_step(self, grad, lr, weight_decay):
        """one torch.optim.Adam step (coupled weight decay, trainer.py:121-122) from a gradient in packed order"""
        grad = self._to_native_dev(grad)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.nnest_spline_adam_step(self._h, _lib.ptr(grad), ctypes.c_float(lr), ctypes.c_float(weight_decay),
                                                        _lib.current_stream(self.device)))

    def train_epochs(self, xtrain, xvalid, perm, noise=None, seed=0, jitter=0.0, batch=100, max_epochs=1, patience=50,
                     lr=1e-3, weight_decay=1e-6, epoch_offset=0, resume=False, finalize=True, result=None):
        """Trainer.train's epoch loop (trainer.py:198-241); same arguments and return value as HipNVP.train_epochs.  The
        best-validation weights are restored on return."""
        assert not resume and epoch_offset == 0, 'the spline trainer takes a run in one call'
        dev = self.device
        if int(batch) > TRAIN_KERNEL_MAX_BATCH:   # the reference takes any batch_size (trainer.py:36, :76, :185): a slower path, not a refusal
            return train_epochs_host(self, xtrain, xvalid, perm, noise, seed, jitter, batch, max_epochs, patience, lr, weight_decay,
                                     chunk_rows=1 << 30)   # (nnest_spline_loss_grad takes a minibatch of any size in one call)
        xtrain = _as_dev_f32(xtrain, dev)
        xvalid = _as_dev_f32(xvalid, dev)
        perm = perm.to(device=dev, dtype=torch.int32).contiguous()
        n_train, n_valid = xtrain.shape[0], xvalid.shape[0]
        assert perm.numel() == max_epochs * n_train
        if noise is not None:
            noise = noise.to(device=dev, dtype=torch.float32).contiguous()
            assert noise.numel() == max_epochs * n_train * self.D
        if not self.data_dep_init_done and max_epochs > 0:
            m = min(int(batch), n_train)
            first = xtrain[perm.view(max_epochs, n_train)[0, :m].long()]
            if noise is not None:
                nz = noise.view(max_epochs, n_train, self.D)[0, :m]
            else:   # drawn from `seed`, not from the global generator: ranks that share the seed initialise identical replicas
                g = torch.Generator(device=dev)
                g.manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
                nz = torch.randn(m, self.D, device=dev, generator=g)
            self.actnorm_init(first + float(jitter) * nz)
        losses = np.zeros((max(max_epochs, 1), 2), np.float32)
        res = _lib.TrainResult()
        with torch.cuda.device(dev):
            _lib.check(self._lib.nnest_spline_train(self._h, _lib.ptr(xtrain), n_train, _lib.ptr(xvalid), n_valid, _lib.ptr(perm),
                                                    _lib.ptr(noise), int(seed) & 0xFFFFFFFFFFFFFFFF, float(jitter), int(batch),
                                                    int(max_epochs), int(patience), float(lr), float(weight_decay),
                                                    losses.ctypes.data_as(ctypes.c_void_p), ctypes.byref(res),
                                                    _lib.current_stream(dev)))
        return dict(losses=torch.from_numpy(losses), epochs_run=res.epochs_run, best_epoch=res.best_epoch,
                    best_validation_loss=res.best_validation_loss, last_train_loss=res.last_train_loss, counter=res.counter,
                    stopped=bool(res.stopped), result=None)
