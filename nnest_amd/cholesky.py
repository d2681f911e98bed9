"""HipCholesky: the reference's 'choleksy' flow (SingleSpeedCholeksy, nnest/networks.py:162-239) -- one linear map
y = L x + b with L lower triangular -- on the nnest_chol_* entry points.  No fused proposal kernel: the sampler drives it
through the host protocol."""
import ctypes

import numpy as np
import torch

from . import _lib
from .flow import _HipFlow, _as_dev_f32


class HipCholesky(_HipFlow):

    def __init__(self, num_inputs, device=None, seed=None):
        if not torch.cuda.is_available():
            raise _lib.NnestHipError('HipCholesky needs an MI355X visible to PyTorch-ROCm; there is no CPU fallback')
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.D = self.num_inputs = int(num_inputs)
        self._lib = _lib.load()
        L = self._lib
        self._sym = dict(forward=L.nnest_chol_forward, inverse=L.nnest_chol_inverse, log_probs=L.nnest_chol_log_probs,
                         set_base=L.nnest_chol_set_base)
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(L.nnest_chol_create(self.D, ctypes.byref(self._h)))
        self.num_params = L.nnest_chol_num_params(self._h)
        self.prior = torch.distributions.MultivariateNormal(torch.zeros(self.D, device=self.device),
                                                            torch.eye(self.D, device=self.device))
        self.load_packed(self.default_init())

    def __del__(self):
        try:
            if getattr(self, '_h', None) is not None and self._h.value:
                self._lib.nnest_chol_destroy(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass

    def layer_shapes(self):
        D = self.D
        return [('flow.flows.0.bias', (D,)), ('flow.flows.0.lower_entries', (D * (D - 1) // 2,)), ('flow.flows.0.unconstrained_diag', (D,))]

    def default_init(self):
        """identity_init (networks.py:183-189): bias 0, lower 0, unconstrained_diag = log(e^{1 - eps} - 1) so that diag = 1"""
        D = self.D
        w = np.zeros(self.num_params, np.float32)
        w[-D:] = np.log(np.exp(1 - 1e-3) - 1)
        return w

    def load_packed(self, packed):
        packed = np.ascontiguousarray(packed, dtype=np.float32)
        if packed.size != self.num_params:
            raise ValueError('expected %d packed weights, got %d' % (self.num_params, packed.size))
        with torch.cuda.device(self.device):
            _lib.check(self._lib.nnest_chol_load_weights(self._h, packed.ctypes.data_as(ctypes.c_void_p), _lib.current_stream(self.device)))

    def store_packed(self):
        out = np.empty(self.num_params, np.float32)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.nnest_chol_store_weights(self._h, out.ctypes.data_as(ctypes.c_void_p), _lib.current_stream(self.device)))
        return out

    def state_dict(self):
        packed, sd, off = self.store_packed(), {}, 0
        for name, shape in self.layer_shapes():
            n = int(np.prod(shape))
            sd[name] = torch.from_numpy(packed[off:off + n].reshape(shape).copy())
            off += n
        return sd

    def load_state_dict(self, sd):
        self.load_packed(np.concatenate([np.asarray(sd[n].detach().cpu().numpy() if torch.is_tensor(sd[n]) else sd[n], dtype=np.float32).ravel()
                                         for n, _ in self.layer_shapes()]))

    # the fused kernels do not know this flow: the sampler falls back to the host protocol
    mh_steps = None
    inverse_loglike = None

    epoch_chunk = 1 << 30

    def loss_grad(self, x):
        x = _as_dev_f32(x, self.device)
        grad = torch.empty(self.num_params, dtype=torch.float32, device=self.device)
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.nnest_chol_loss_grad(self._h, _lib.ptr(x), x.shape[0], _lib.ptr(grad), _lib.ptr(loss),
                                                      _lib.current_stream(self.device)))
        return loss, grad

    def train_epochs(self, xtrain, xvalid, perm, noise=None, seed=0, jitter=0.0, batch=100, max_epochs=1, patience=50,
                     lr=1e-3, weight_decay=1e-6, epoch_offset=0, resume=False, finalize=True, result=None):
        """Trainer.train's epoch loop (trainer.py:198-241), host-driven; arguments and return value as HipNVP.train_epochs"""
        assert not resume and epoch_offset == 0
        dev = self.device
        xtrain = _as_dev_f32(xtrain, dev)
        xvalid = _as_dev_f32(xvalid, dev)
        n_train, n_valid = xtrain.shape[0], xvalid.shape[0]
        perm = perm.to(device=dev, dtype=torch.int64).view(max_epochs, n_train)
        if noise is not None:
            noise = noise.to(device=dev, dtype=torch.float32).view(max_epochs, n_train, self.D)
        gen = torch.Generator(device=dev)
        gen.manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
        losses = np.zeros((max(max_epochs, 1), 2), np.float32)
        best, best_epoch, counter, stopped, epochs_run = float('inf'), 0, 0, False, 0
        best_w = self.store_packed()
        for epoch in range(max_epochs):
            step_losses = []
            for b0 in range(0, n_train, batch):
                rows = xtrain[perm[epoch, b0:b0 + batch]]
                if jitter != 0.0:
                    nz = noise[epoch, b0:b0 + batch] if noise is not None else torch.randn(rows.shape, device=dev, generator=gen)
                    rows = rows + float(jitter) * nz
                loss, grad = self.loss_grad(rows.contiguous())
                with torch.cuda.device(dev):
                    _lib.check(self._lib.nnest_chol_adam_step(self._h, _lib.ptr(grad), ctypes.c_float(lr), ctypes.c_float(weight_decay),
                                                              _lib.current_stream(dev)))
                step_losses.append(loss)
            train_loss = float(torch.stack(step_losses).sum()) / n_train
            valid_loss = float(-self.log_probs(xvalid).mean()) / n_valid
            losses[epoch] = (train_loss, valid_loss)
            epochs_run = epoch + 1
            if valid_loss < best:
                best, best_epoch, counter, best_w = valid_loss, epoch + 1, 0, self.store_packed()
            counter += 1
            if counter > patience:
                stopped = True
                break
        self.load_packed(best_w)
        return dict(losses=torch.from_numpy(losses), epochs_run=epochs_run, best_epoch=best_epoch, best_validation_loss=best,
                    last_train_loss=float(losses[max(epochs_run - 1, 0), 0]), counter=counter, stopped=stopped, result=None)
