"""HipMAF: a masked autoregressive flow resident on one MI355X, driven through the C ABI (nnest_maf_create; every other call
is the nnest_nvp_* entry point of the same name -- the handle is an nnest_nvp_t).

ABSENT FROM THE REFERENCE: nnest/trainer.py:83-100 builds 'choleksy', 'nvp' or 'spline' only.  BASELINE.json's north star and
its config 5 name a MAF, so this build defines one (nnest_amd/csrc/maf_tile.h) behind the reference's flow protocol
(nnest/networks.py:17-84: forward, inverse, log_probs, sample) and its Trainer seam (`Trainer(flow='maf')`):
B blocks of two MADE-masked nets with the shapes of the reference's coupling nets, order reversed between blocks; forward
(density / training) is one pass, inverse (sampling, MCMC proposals) runs group by group.  Parity is against a CPU
restatement of the same definition (tests/) and against self-consistency (round trip, log-det), as the reference tests its own flows
(tests/test_flows.py:27-30)."""
import ctypes

import numpy as np
import torch

from . import _lib
from .flow import HipNVP, _as_dev_f32


class HipMAF(HipNVP):
    """num_inputs=D, num_hidden=H (16), num_blocks=B, num_layers=L.  The state_dict has the keys and shapes of SingleSpeedNVP
    (scale_net / translate_net per block); masked entries are kept (they are zero in effect and take only Adam's weight-decay
    steps, like the entries RealNVP's mask never reaches)."""

    epoch_chunk = 1 << 30   # Trainer.train hands the whole run to train_epochs (the loop below keeps the early-stopping books)

    def __init__(self, num_inputs, num_hidden=16, num_blocks=3, num_layers=1, device=None, seed=None):
        if not torch.cuda.is_available():
            raise _lib.NnestHipError('HipMAF needs an MI355X visible to PyTorch-ROCm (torch.cuda.is_available() is False); '
                                     'there is no CPU fallback')
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.D, self.H, self.B, self.L = int(num_inputs), int(num_hidden), int(num_blocks), int(num_layers)
        self.num_inputs = self.D
        self.scale = ''
        self._lib = _lib.load()
        L = self._lib
        self._sym = dict(forward=L.nnest_nvp_forward, inverse=L.nnest_nvp_inverse, log_probs=L.nnest_nvp_log_probs,
                         inverse_loglike=L.nnest_nvp_inverse_loglike, mh=L.nnest_mh_constrained_steps, set_base=L.nnest_nvp_set_base)
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(L.nnest_maf_create(self.D, self.H, self.B, self.L, ctypes.byref(self._h)))
        self.num_params = L.nnest_nvp_num_params(self._h)
        self._set_pad_index(None, self.num_params)
        self.num_groups = L.nnest_maf_num_groups(self._h)   # passes of the nets per block in the sampling direction
        self.prior = torch.distributions.MultivariateNormal(torch.zeros(self.D, device=self.device),
                                                            torch.eye(self.D, device=self.device))
        self.load_packed(self.default_init(seed))

    def train_epochs(self, xtrain, xvalid, perm, noise=None, seed=0, jitter=0.0, batch=100, max_epochs=1, patience=50,
                     lr=1e-3, weight_decay=1e-6, epoch_offset=0, resume=False, finalize=True, result=None, one_cu=False):
        """Trainer.train's epoch loop (trainer.py:198-241) driven from the host: per minibatch one gradient (nnest_nvp_loss_grad:
        two launches) and one Adam step + image rebuild, an epoch's minibatches queued by one call (nnest_maf_train_epoch);
        arguments and return value as HipNVP.train_epochs"""
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
            tot = torch.zeros((), dtype=torch.float32, device=dev)
            # the epoch's rows in minibatch order, jitter applied (data + jitter * randn, trainer.py:392): one gather and one
            # draw per epoch; a minibatch is a slice (nothing is queued between its two library calls but the loss add)
            rows_all = xtrain[perm[epoch]]
            if jitter != 0.0:
                nz = noise[epoch] if noise is not None else torch.randn(rows_all.shape, device=dev, generator=gen)
                rows_all = rows_all + float(jitter) * nz
            rows_all = rows_all.contiguous()
            with torch.cuda.device(dev):   # every minibatch of the epoch queued by one library call (nothing read back)
                _lib.check(self._lib.nnest_maf_train_epoch(self._h, _lib.ptr(rows_all), n_train, int(batch), ctypes.c_float(lr),
                                                           ctypes.c_float(weight_decay), _lib.ptr(tot), _lib.current_stream(dev)))
            # Trainer._validate (trainer.py:405-418): the SUM over minibatches of each minibatch's mean, / len(dataset) -- not the
            # mean over the whole set (they differ by the number of minibatches, and in detail when the last one is ragged)
            lp = self.log_probs(xvalid)
            vsum = torch.stack([-c.mean() for c in lp.split(int(batch))]).sum()
            both = torch.stack([tot, vsum.to(tot.dtype)]).cpu()             # one read-back per epoch
            train_loss = float(both[0]) / n_train                           # trainer.py:403
            valid_loss = float(both[1]) / n_valid
            losses[epoch] = (train_loss, valid_loss)
            epochs_run = epoch + 1
            if valid_loss < best:                                           # trainer.py:205-209
                best, best_epoch, counter, best_w = valid_loss, epoch + 1, 0, self.store_packed()
            counter += 1
            if counter > patience:                                          # trainer.py:223-232
                stopped = True
                break
        self.load_packed(best_w)                                            # netG.load_state_dict(best_model)  trainer.py:241
        return dict(losses=torch.from_numpy(losses), epochs_run=epochs_run, best_epoch=best_epoch, best_validation_loss=best,
                    last_train_loss=float(losses[max(epochs_run - 1, 0), 0]), counter=counter, stopped=stopped, result=None)
