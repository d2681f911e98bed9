"""HipFastSlowNVP: the reference's fast/slow hierarchy for the RealNVP flow (FastSlowNVP, nnest/networks.py:86-150, :350-380;
`Trainer(num_slow=...)`, trainer.py:85-88) on the HIP kernels.

Model: a slow NVP on the first `num_slow` dimensions, a fast NVP on the rest, then ONE coupling layer (hidden 64, one
hidden layer, networks.py:112-120) that transforms the fast block conditioned on the slow block -- so a move of the fast
latent coordinates alone leaves the slow physical coordinates exactly unchanged (tests/test_flows.py:94-118).

Composition here: three `HipNVP` handles.  The final coupling has a block mask (slow | fast) while the kernels' masks
alternate by parity, so it runs as a one-block NVP on an interleaved vector of length 2 max(S, F): odd positions carry the
slow coordinates (conditioning side of block 0), even positions the fast ones, the shorter side zero-padded with
zero, frozen weights.  Training is the chain rule over the three stages: `nnest_nvp_vjp` per stage, `nnest_nvp_adam_step`
per stage (torch.optim.Adam over all parameters of the model, trainer.py:121-122).  Tensor slicing / interleaving between
the stages is done with torch indexing (plumbing); there is no fused proposal kernel for this model -- the sampler drives
it through the host protocol (`Sampler._mcmc_sample_host`), as the reference does."""
import ctypes

import numpy as np
import torch

from . import _lib
from .flow import HipNVP, _as_dev_f32


class HipFastSlowNVP(object):
    kind = 'nvp'

    def _make_stages(self, device, seeds):
        fast = HipNVP(self.F, self.H, self.B, self.L, device=device, seed=seeds[0])
        slow = HipNVP(self.S, self.H, self.B, self.L, device=device, seed=seeds[1])
        return fast, slow

    def __init__(self, num_fast, num_slow, num_hidden=16, num_blocks=3, num_layers=1, device=None, seed=None):
        self.F, self.S = int(num_fast), int(num_slow)
        self.D = self.num_inputs = self.F + self.S
        self.H, self.B, self.L = int(num_hidden), int(num_blocks), int(num_layers)
        self.m = max(self.F, self.S)
        if self.m > 16:
            raise _lib.NnestHipError('fast/slow hierarchy: max(num_slow, num_fast) = %d > 16 (the hidden-64 coupling kernel is '
                                     'instantiated for 32 interleaved dimensions)' % self.m)
        seeds = [None] * 3 if seed is None else [int(seed), int(seed) + 1, int(seed) + 2]
        self.fast, self.slow = self._make_stages(device, seeds)
        self.coupling = HipNVP(2 * self.m, 64, 1, 1, device=device, seed=seeds[2])
        self.device = self.fast.device
        self._lib = self.fast._lib
        self.prior = torch.distributions.MultivariateNormal(torch.zeros(self.D, device=self.device),
                                                            torch.eye(self.D, device=self.device))
        # the reference's coupling parameters that the masks never use (input columns of the fast block, output rows of
        # the slow block): kept as loaded so that state_dict() round-trips
        self._unused = {}
        self._cmask = torch.from_numpy(self._coupling_mask()).to(self.device)
        self._load_coupling(self._coupling_reference_init(self.coupling.store_packed()))
        self.num_params = sum(int(np.prod(s)) for _, s in self.layer_shapes())

    # ---- the interleaved coupling ---------------------------------------------------------------------------------
    def _coupling_shapes(self):
        H = 64
        return [('0.weight', (H, self.D)), ('0.bias', (H,)), ('2.weight', (H, H)), ('2.bias', (H,)), ('4.weight', (self.D, H)),
                ('4.bias', (self.D,))]

    def _coupling_to_packed(self, ref):
        """reference tensors {net: {leaf: array}} -> packed vector of the 1-block NVP on 2m interleaved dims"""
        S, F, m, H = self.S, self.F, self.m, 64
        out = []
        for net in ('scale_net', 'translate_net'):
            W0 = np.zeros((H, 2 * m), np.float32)
            W0[:, 1:2 * S:2] = ref[net]['0.weight'][:, :S]
            Wo = np.zeros((2 * m, H), np.float32)
            Wo[0:2 * F:2, :] = ref[net]['4.weight'][S:, :]
            bo = np.zeros(2 * m, np.float32)
            bo[0:2 * F:2] = ref[net]['4.bias'][S:]
            out += [W0.ravel(), ref[net]['0.bias'], ref[net]['2.weight'].ravel(), ref[net]['2.bias'], Wo.ravel(), bo]
        return np.concatenate(out).astype(np.float32)

    def _coupling_from_packed(self, packed):
        S, F, m, H = self.S, self.F, self.m, 64
        ref, off = {}, 0
        for net in ('scale_net', 'translate_net'):
            sizes = [H * 2 * m, H, H * H, H, 2 * m * H, 2 * m]
            parts = []
            for n in sizes:
                parts.append(packed[off:off + n])
                off += n
            un = self._unused.get(net, {})
            W0 = np.array(un.get('0.weight', np.zeros((H, self.D), np.float32)), copy=True)
            W0[:, :S] = parts[0].reshape(H, 2 * m)[:, 1:2 * S:2]
            Wo = np.array(un.get('4.weight', np.zeros((self.D, H), np.float32)), copy=True)
            Wo[S:, :] = parts[4].reshape(2 * m, H)[0:2 * F:2, :]
            bo = np.array(un.get('4.bias', np.zeros(self.D, np.float32)), copy=True)
            bo[S:] = parts[5][0:2 * F:2]
            ref[net] = {'0.weight': W0, '0.bias': parts[1].copy(), '2.weight': parts[2].reshape(H, H).copy(), '2.bias': parts[3].copy(),
                        '4.weight': Wo, '4.bias': bo}
        return ref

    def _coupling_mask(self):
        """1 for the packed entries of the interleaved coupling that are real parameters, 0 for the zero padding (first-layer
        columns and last-layer rows / biases at positions where no real dimension sits)"""
        ones = {net: {leaf: np.ones(shape, np.float32) for leaf, shape in self._coupling_shapes()} for net in ('scale_net', 'translate_net')}
        return self._coupling_to_packed(ones)

    def _coupling_reference_init(self, packed_default):
        """nn.Linear default init of CouplingLayer(D, 64, ...) (networks.py:112-120): drawn through a throw-away packed vector
        of the interleaved model (same distribution family: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) with fan_in = D, not 2m)"""
        H, D = 64, self.D
        g = torch.Generator()
        g.manual_seed(int(np.abs(packed_default[:8]).sum() * 1e6) % (2 ** 31))
        ref = {}
        for net in ('scale_net', 'translate_net'):
            ref[net] = {}
            for leaf, shape in self._coupling_shapes():
                fan_in = shape[1] if len(shape) == 2 else last
                last = fan_in
                ref[net][leaf] = ((torch.rand(shape, generator=g) * 2 - 1) / np.sqrt(fan_in)).numpy().astype(np.float32)
        return ref

    def _load_coupling(self, ref):
        self._unused = {net: {k: np.array(v, copy=True) for k, v in ref[net].items() if k in ('0.weight', '4.weight', '4.bias')}
                        for net in ref}
        self.coupling.load_packed(self._coupling_to_packed(ref))

    def _interleave(self, slow, fast):
        v = torch.zeros(slow.shape[0], 2 * self.m, dtype=torch.float32, device=self.device)
        v[:, 1:2 * self.S:2] = slow
        v[:, 0:2 * self.F:2] = fast
        return v

    def _split(self, v):
        return v[:, 1:2 * self.S:2].contiguous(), v[:, 0:2 * self.F:2].contiguous()

    # ---- weights (reference state_dict order: fast_flow, slow_flow, flow) -----------------------------------------------
    def layer_shapes(self):
        out = [(t[0].replace('flow.flows', 'fast_flow.flows', 1), t[1]) for t in self.fast.layer_shapes()]
        out += [(t[0].replace('flow.flows', 'slow_flow.flows', 1), t[1]) for t in self.slow.layer_shapes()]
        for net in ('scale_net', 'translate_net'):
            out += [('flow.flows.0.%s.%s' % (net, leaf), shape) for leaf, shape in self._coupling_shapes()]
        return out

    def state_dict(self):
        sd = {}
        for name, v in self.fast.state_dict().items():
            sd[name.replace('flow.flows', 'fast_flow.flows', 1)] = v
        for name, v in self.slow.state_dict().items():
            sd[name.replace('flow.flows', 'slow_flow.flows', 1)] = v
        ref = self._coupling_from_packed(self.coupling.store_packed())
        for net in ('scale_net', 'translate_net'):
            for leaf, shape in self._coupling_shapes():
                sd['flow.flows.0.%s.%s' % (net, leaf)] = torch.from_numpy(np.asarray(ref[net][leaf], np.float32).reshape(shape).copy())
        return sd

    def load_state_dict(self, sd, P=None):
        def arr(v):
            return np.asarray(v.detach().cpu().numpy() if torch.is_tensor(v) else v, dtype=np.float32)
        extra = ({}, {}) if P is None else ({'P': P['fast']}, {'P': P['slow']})   # spline stages: the 1x1 convs' permutations
        self.fast.load_state_dict({k.replace('fast_flow.flows', 'flow.flows', 1): v for k, v in sd.items() if k.startswith('fast_flow.')},
                                  **extra[0])
        self.slow.load_state_dict({k.replace('slow_flow.flows', 'flow.flows', 1): v for k, v in sd.items() if k.startswith('slow_flow.')},
                                  **extra[1])
        ref = {net: {leaf: arr(sd['flow.flows.0.%s.%s' % (net, leaf)]) for leaf, _ in self._coupling_shapes()}
               for net in ('scale_net', 'translate_net')}
        self._load_coupling(ref)

    def store_packed(self):
        """the concatenated reference state_dict"""
        return np.concatenate([v.numpy().ravel() for v in self.state_dict().values()]).astype(np.float32)

    def load_packed(self, packed, P=None):
        packed = np.asarray(packed, dtype=np.float32)
        sd, off = {}, 0
        for name, shape in self.layer_shapes():
            n = int(np.prod(shape))
            sd[name] = packed[off:off + n].reshape(shape)
            off += n
        if off != packed.size:
            raise ValueError('expected %d packed weights, got %d' % (off, packed.size))
        self.load_state_dict(sd, P)

    def used_mask(self):
        """True for the entries of store_packed() that the model's output depends on (the masks never reach the rest)"""
        out = []
        for name, shape in self.layer_shapes():
            mk = np.ones(shape, bool)
            if name.startswith('flow.flows.0.'):
                leaf = '.'.join(name.split('.')[-2:])
                if leaf == '0.weight':
                    mk[:, self.S:] = False
                elif leaf in ('4.weight', '4.bias'):
                    mk[:self.S] = False
            out.append(mk.ravel())
        return np.concatenate(out)

    def eval(self):
        return self

    def train(self, mode=True):
        return self

    def parameters(self):
        return list(self.state_dict().values())

    # ---- passes (networks.py:125-138) -----------------------------------------------------------------------------------
    def forward(self, x):
        x = _as_dev_f32(x, self.device)
        zs, lds = self.slow.forward(x[:, :self.S].contiguous())
        zf, ldf = self.fast.forward(x[:, self.S:].contiguous())
        v, ldc = self.coupling.forward(self._interleave(zs, zf))
        s, f = self._split(v)
        return torch.cat([s, f], dim=1), lds + ldf + ldc

    def inverse(self, z):
        z = _as_dev_f32(z, self.device)
        v, ldc = self.coupling.inverse(self._interleave(z[:, :self.S], z[:, self.S:]))
        ys, yf = self._split(v)
        xs, lds = self.slow.inverse(ys)
        xf, ldf = self.fast.inverse(yf)
        return torch.cat([xs, xf], dim=1), ldc + lds + ldf

    def log_probs(self, x):
        """networks.py:140-146: prior.log_prob(u) + all three log-dets; the interleaved coupling's own log_probs carries the
        N(0, I) density of its 2m coordinates, of which the 2m - D padding ones are exactly 0"""
        x = _as_dev_f32(x, self.device)
        zs, lds = self.slow.forward(x[:, :self.S].contiguous())
        zf, ldf = self.fast.forward(x[:, self.S:].contiguous())
        lp = self.coupling.log_probs(self._interleave(zs, zf))
        return lp + (2 * self.m - self.D) * 0.9189385332046727 + lds + ldf

    def sample(self, num_samples=None, noise=None):
        if noise is None:
            noise = torch.randn(num_samples, self.D, device=self.device)
        x, _ = self.inverse(noise)
        return x

    def prior_sample(self, num_samples):
        return torch.randn(int(num_samples), self.D, device=self.device)

    # ---- training ---------------------------------------------------------------------------------------------------------
    epoch_chunk = 1 << 30

    def _vjp(self, net, x, gz, gld):
        return net.vjp(x, gz, gld)

    def _adam(self, net, grad, lr, wd):
        net.adam_step(grad, lr, wd)

    def loss_grad(self, x):
        """loss = -mean(log_probs(x)) and its gradient in the three stages' packed layouts (fast, slow, coupling)"""
        x = _as_dev_f32(x, self.device)
        M = x.shape[0]
        xs, xf = x[:, :self.S].contiguous(), x[:, self.S:].contiguous()
        zs, lds = self.slow.forward(xs)
        zf, ldf = self.fast.forward(xf)
        v = self._interleave(zs, zf)
        lp = self.coupling.log_probs(v) + (2 * self.m - self.D) * 0.9189385332046727 + lds + ldf
        z, _ = self.coupling.forward(v)
        gC, gv = self._vjp(self.coupling, v, z / M, -1.0 / M)      # d(-mean log N(z))/dz = z / M
        gC = gC * self._cmask
        gys, gyf = self._split(gv)
        gS, _ = self._vjp(self.slow, xs, gys, -1.0 / M)
        gF, _ = self._vjp(self.fast, xf, gyf, -1.0 / M)
        return -lp.mean(), (gF, gS, gC)

    def reference_gradient(self, grads):
        """(gF, gS, gC) -> the gradient in the order of store_packed() (zero where the masks never reach)"""
        gF, gS, gC = [g.detach().cpu().numpy() for g in grads]
        ref = self._coupling_from_packed_plain(gC)
        parts = [gF, gS]
        for net in ('scale_net', 'translate_net'):
            parts += [np.asarray(ref[net][leaf], np.float32).ravel() for leaf, _ in self._coupling_shapes()]
        return np.concatenate(parts)

    def _coupling_from_packed_plain(self, packed):
        saved = self._unused
        self._unused = {}
        try:
            return self._coupling_from_packed(packed)
        finally:
            self._unused = saved

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
        keep_P = getattr(self, 'P', None)
        for epoch in range(max_epochs):
            tot = 0.0
            for b0 in range(0, n_train, batch):
                idx = perm[epoch, b0:b0 + batch]
                rows = xtrain[idx]
                if jitter != 0.0:
                    nz = noise[epoch, b0:b0 + batch] if noise is not None else torch.randn(rows.shape, device=dev, generator=gen)
                    rows = rows + float(jitter) * nz
                loss, (gF, gS, gC) = self.loss_grad(rows)
                self._adam(self.fast, gF, lr, weight_decay)
                self._adam(self.slow, gS, lr, weight_decay)
                self._adam(self.coupling, gC, lr, weight_decay)
                tot += float(loss)
            train_loss = tot / n_train
            valid_loss = float(-self.log_probs(xvalid).mean()) / n_valid
            losses[epoch] = (train_loss, valid_loss)
            epochs_run = epoch + 1
            if valid_loss < best:
                best, best_epoch, counter, best_w = valid_loss, epoch + 1, 0, self.store_packed()
            counter += 1
            if counter > patience:
                stopped = True
                break
        self.load_packed(best_w, keep_P)
        if keep_P is not None:
            self.fast.data_dep_init_done = self.slow.data_dep_init_done = True
        return dict(losses=torch.from_numpy(losses), epochs_run=epochs_run, best_epoch=best_epoch, best_validation_loss=best,
                    last_train_loss=float(losses[max(epochs_run - 1, 0), 0]), counter=counter, stopped=stopped, result=None)


class HipFastSlowSpline(HipFastSlowNVP):
    """FastSlowSpline (nnest/networks.py:718-731): as above with neural-spline stages -- the fast stage always with hidden
    width 16 (networks.py:722), the slow stage with `hidden_dim` -- and the same hidden-64 NVP coupling on top."""
    kind = 'spline'

    def _make_stages(self, device, seeds):
        from .spline import HipSpline
        fast = HipSpline(self.F, 16, self.B, device=device, seed=seeds[0])
        slow = HipSpline(self.S, self.H, self.B, device=device, seed=seeds[1])
        return fast, slow

    def __init__(self, num_fast, num_slow, hidden_dim=16, num_blocks=3, device=None, seed=None):
        if num_fast < 2 or num_slow < 2:
            raise ValueError('NSF_CL needs at least 2 dimensions per block (networks.py:565-574)')
        super().__init__(num_fast, num_slow, hidden_dim, num_blocks, 1, device=device, seed=seed)

    @property
    def P(self):
        return {'fast': self.fast.P, 'slow': self.slow.P}

    @property
    def data_dep_init_done(self):
        return self.fast.data_dep_init_done and self.slow.data_dep_init_done

    @data_dep_init_done.setter
    def data_dep_init_done(self, value):
        self.fast.data_dep_init_done = self.slow.data_dep_init_done = bool(value)
