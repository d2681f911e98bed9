"""NestedSampler: the reference's nested-sampling outer loop (nnest/nested.py:24-510) kept on the host in
float64, driving the HIP hot path: flow retrain (Trainer.train -> nnest_nvp_train) and batched constrained
Metropolis proposals (Sampler._mcmc_sample -> nnest_mh_constrained_steps).

Same constructor and run() keywords, same result attributes (logz, samples, weights, loglikes) and the same
on-disk products (results/results.csv, results/final.csv, checkpoint/*.npy + checkpoint_<it>.txt,
chains/chain.txt) as the reference.  Strategies: 'rejection_prior' and 'mcmc' (the reference default pair,
nested.py:136-137), and the low-dimensional 'rejection_flow' / 'density_flow' (nested.py:336-360), which examine
a block of candidates per flow launch instead of one.

Multi-GPU (torch.distributed, one process per GPU): the live set and the evidence state are replicated on
every rank; rank 0 draws all host-side randomness and broadcasts it; each MCMC batch is sharded over ranks
and its endpoints are all-gathered on device memory (RCCL); every rank retrains its own replica of the flow from
one broadcast seed -- the training kernels are bitwise reproducible, so the replicas stay identical with no weight
broadcast (a trainer that cannot promise that is trained on rank 0 and its weights are broadcast).

Checkpoints (nested.py:473-485: every `log_interval` accepted points the reference rewrites every dead point and chain.txt as
text -- O(n^2) over a run, 995 s of a 15 s GPU run).  Here a checkpoint writes the live set and APPENDS the new dead points to
saved_*.npy (GrowingNpy: the same files, valid .npy at any time), so it costs ~1 ms; it is taken at the reference's cadence but
at most once per `checkpoint_min_seconds` (default 2; 0 = exactly the reference's cadence and file set), and chain.txt -- a
derived product -- is rewritten at most once per `chain_min_seconds` (default 30) and at the end.
"""
import csv
import glob
import json
import logging
import os
import time

import numpy as np
import torch

import ctypes

from . import _lib
from .priors import UniformPrior
from .sampler import Sampler
from .utils import GrowingNpy


class _Evidence(object):
    """Running evidence log Z and information H as dead points arrive with weight logwt = log(volume shell) + logL
    (the recurrence of nested.py:280-284, applied to the final live points in nested.py:487-495 too)."""

    def __init__(self, logz=-1e300, h=0.0):
        self.logz, self.h = logz, h

    def add(self, logwt, logl):
        total = np.logaddexp(self.logz, logwt)
        self.h = np.exp(logwt - total) * logl + np.exp(self.logz - total) * (self.h + self.logz) - total
        self.logz = total


def _first_live(strategy, expired):
    """the sampling method in force: the first of `strategy` that has not expired (nested.py:300-306)"""
    return next((m for m in strategy if m not in expired), '')


class NestedSampler(Sampler):

    def __init__(self,
                 x_dim,
                 loglike,
                 transform=None,
                 append_run_num=True,
                 hidden_dim=16,
                 num_slow=0,
                 num_derived=0,
                 batch_size=100,
                 flow='spline',
                 num_blocks=3,
                 num_layers=1,
                 learning_rate=0.001,
                 log_dir='logs/test',
                 resume=True,
                 base_dist=None,
                 scale='',
                 use_gpu=True,
                 trainer=None,
                 oversample_rate=-1,
                 log_level=logging.INFO,
                 param_names=None,
                 num_live_points=1000,
                 fused=True,
                 mcmc_history=False,
                 checkpoint_min_seconds=2.0,
                 chain_min_seconds=30.0,
                 native_loop=True,
                 mcmc_proposal='mh'):
        prior = UniformPrior(x_dim, -1, 1)  # nested.py:76
        super(NestedSampler, self).__init__(x_dim, loglike, transform=transform, append_run_num=append_run_num,
                                            hidden_dim=hidden_dim, num_slow=num_slow, num_derived=num_derived,
                                            batch_size=batch_size, flow=flow, num_blocks=num_blocks,
                                            num_layers=num_layers, learning_rate=learning_rate, log_dir=log_dir,
                                            resume=resume, use_gpu=use_gpu, base_dist=base_dist, scale=scale,
                                            trainer=trainer, prior=prior, transform_prior=False, log_level=log_level,
                                            param_names=param_names, oversample_rate=oversample_rate, fused=fused,
                                            mcmc_history=mcmc_history, mcmc_proposal=mcmc_proposal)
        self.num_live_points = num_live_points
        self.checkpoint_min_seconds = checkpoint_min_seconds
        self.chain_min_seconds = chain_min_seconds
        # native_loop (not in the reference): the per-iteration body of run() under the 'mcmc' strategy in the native library
        # (nnest_host_mcmc_consume); False = the Python restatement of the same loop below (kept as the readable form and the
        # cross-check: tests/test_host_nested.py runs both and requires equal trajectories)
        self.native_loop = native_loop
        self._grow = None
        self.sampler = 'nested'
        if self.single_or_primary_process:
            self.logger.info('Num live points [%d]' % self.num_live_points)
            with open(os.path.join(self.logs['results'], 'results.csv'), 'w') as f:
                csv.writer(f).writerow(['step', 'acceptance', 'min_ess', 'max_ess', 'jump_distance', 'scale', 'loglstar',
                                        'logz', 'fraction_remain', 'ncall'])

    # ---- helpers ---------------------------------------------------------------------------------------------
    def _initial_loglikes(self, active_u):
        """nested.py:210-228: likelihood of the initial live points through the host protocol (float64),
        sharded over ranks and all-gathered when distributed."""
        if not self.use_mpi:
            return self.loglike(active_u)
        N = active_u.shape[0]
        per = -(-N // self.mpi_size)
        rows = np.arange(self.mpi_rank * per, (self.mpi_rank + 1) * per) % N
        logl, derived = self.loglike(active_u[rows])
        logl = self._all_gather_rows(logl)[:N] if per * self.mpi_size == N else self._gather_padded(logl, N, per)
        derived = np.empty((N, self.num_derived)) if self.num_derived == 0 else self._gather_padded(derived, N, per)
        return logl, derived

    def _gather_padded(self, arr, N, per):
        full = self._all_gather_rows(arr)
        out = np.empty((N,) + arr.shape[1:], dtype=arr.dtype)
        for r in range(self.mpi_size):
            rows = np.arange(r * per, (r + 1) * per) % N
            out[rows] = full[r * per:(r + 1) * per]
        return out

    def _train(self, active_u, train_iters, jitter):
        """nested.py:311-314.  Distributed: every rank trains the same replica from one broadcast seed when the trainer is
        bitwise reproducible (`replicable`); otherwise rank 0 trains and the packed weights are broadcast (C3)."""
        if not self.use_mpi:
            self.trainer.train(active_u, max_iters=train_iters, jitter=jitter)
            return
        if getattr(self.trainer, 'replicable', False):
            if not getattr(self, '_replicas_aligned', False):   # once: every rank starts from rank 0's weights
                self._broadcast_weights()
                self._replicas_aligned = True
            seed = np.array([self._next_seed() & 0x7FFFFFFF if self.mpi_rank == 0 else 0], dtype=np.int64)
            seed = int(self._broadcast(seed)[0])
            self.trainer.train(active_u, max_iters=train_iters, jitter=jitter, rng_seed=seed)
            self._guard_replicas()
            return
        if self.mpi_rank == 0:
            self.trainer.train(active_u, max_iters=train_iters, jitter=jitter)
        self._broadcast_weights()

    def _replica_state(self):
        """what a replica is: the packed weights and, where the flow exposes them, Adam's moments and step count"""
        netG = self.trainer.netG
        parts = [np.ascontiguousarray(netG.store_packed(), dtype=np.float32)]
        if hasattr(netG, 'adam_moments') and hasattr(netG, 'set_adam'):
            m, v = netG.adam_moments()
            parts += [np.ascontiguousarray(m, dtype=np.float32), np.ascontiguousarray(v, dtype=np.float32),
                      np.array([netG.adam_step_count()], dtype=np.int64)]
        return parts

    def _guard_replicas(self):
        """The replicas of a replicated retrain are identical because the training kernels are bitwise reproducible -- which
        this checks instead of trusting: a 64-bit checksum of the replica's state (weights + Adam) is gathered over the ranks
        (8 bytes each) after every retrain; on a mismatch the run says so loudly, takes rank 0's weights and Adam state (C3's
        path) and goes on.  A silent divergence would desynchronise the replicated evidence state (nested.py:311-314 trains on
        rank 0 only and never has the question)."""
        import zlib
        parts = self._replica_state()
        h = 0
        for a in parts:
            h = zlib.crc32(a.tobytes(), h)
        h2 = zlib.adler32(b''.join(a.tobytes() for a in parts))
        mine = np.array([(h << 32) | h2], dtype=np.uint64).view(np.int64)
        every = self._all_gather_rows(mine)
        self.replica_checks = getattr(self, 'replica_checks', 0) + 1
        if np.all(every == every[0]):
            return True
        self.replica_repairs = getattr(self, 'replica_repairs', 0) + 1
        self.logger.error('REPLICA DIVERGENCE after retrain %d: checksums %s -- taking rank 0\'s weights and Adam state' %
                          (self.replica_checks, ' '.join('%016x' % int(x) for x in every.view(np.uint64))))
        self._broadcast_weights()
        netG = self.trainer.netG
        if len(parts) == 4:
            m, v, step = self._broadcast(parts[1]), self._broadcast(parts[2]), int(self._broadcast(parts[3])[0])
            if self.mpi_rank != 0:
                netG.set_adam(m, v, step)
        return False

    def _broadcast_weights(self):
        """C3: rank 0's packed weights (<= 86 KB for the NVP shapes) to every rank"""
        netG = self.trainer.netG
        w = self._broadcast(netG.store_packed(), src=0)
        if hasattr(netG, 'P'):  # spline flow: the fixed permutations of the 1x1 convs are not part of the weights
            P = netG.P
            P = {k: self._broadcast(v, src=0) for k, v in P.items()} if isinstance(P, dict) else self._broadcast(P, src=0)
            # ... and whether ActNorm's data-dependent initialisation has happened is a plain attribute (networks.py:698-705)
            done = bool(self._broadcast(np.array([int(netG.data_dep_init_done)], dtype=np.int64))[0])
            if self.mpi_rank != 0:
                netG.load_packed(w, P)
                netG.data_dep_init_done = done
        elif self.mpi_rank != 0:
            netG.load_packed(w)

    def _pinned_form(self, C, dynamic=False):
        """the K4 form the WHOLE batch of C chains would run, pinned on every rank's shard so that the sharded batch
        reproduces the unsharded one bit for bit -- under a fixed step or the per-16-walker rule, which are shard-invariant.  The library is asked (nnest_mh_form_for): which form applies depends on
        the flow's shape (x_dim, hidden_dim, num_blocks, num_layers, scale) and on the step rule as well as on the population."""
        if self.mpi_size == 1 or self._fused_like_id is None:
            return None
        ask = getattr(self.trainer.netG, 'mh_form_for', None)
        if ask is None:
            return None
        mode = 'batch' if dynamic and getattr(self, '_batch_rule_ok', True) else ('group' if dynamic else False)
        if mode == 'batch':
            # the batch-wide rule counts per rank: a sharded batch is not the unsharded one whatever the form,
            # so nothing is pinned and every rank runs the fastest form of its own shard (config 5 on 8 GPUs: the solo form
            # at 1000 walkers per rank instead of the 8000-walker batch's image form, 4x the step rate)
            return None
        lag = getattr(self, 'mcmc_step_lag', None)
        return ask(C, dynamic=mode, lag=lag) or ask(C, dynamic='group' if dynamic else False)

    def _checkpoint(self, it, active_u, active_v, active_logl, active_derived, saved_v, saved_logl, saved_logwt, state):
        cp = self.logs['checkpoint']
        # models/netG.pt as of this checkpoint (the trainer batches its file writes and a worker thread does them): handed over AND
        # waited for, so that a resume never pairs these live points with an older flow
        if hasattr(self.trainer, 'wait_for_saves'):
            self.trainer.wait_for_saves()
        elif hasattr(self.trainer, 'flush_pending_files'):
            self.trainer.flush_pending_files()
        np.save(os.path.join(cp, 'active_u_%s.npy' % it), active_u)
        np.save(os.path.join(cp, 'active_v_%s.npy' % it), active_v)
        np.save(os.path.join(cp, 'active_logl_%s.npy' % it), active_logl)
        np.save(os.path.join(cp, 'active_derived_%s.npy' % it), active_derived)
        # the dead points so far: the same three files as the reference (nested.py:479-481), grown by the new rows only
        if getattr(self, '_grow', None) is None:
            # (created with the rows there are -- a resumed run's dead points included -- through a temporary file that replaces
            # the old one: the rows on disk are never gone; run() drops the writers, so every run starts its own files)
            self._grow = {'saved_v': GrowingNpy(os.path.join(cp, 'saved_v.npy'), (self.x_dim + self.num_derived,), initial=saved_v),
                          'saved_logl': GrowingNpy(os.path.join(cp, 'saved_logl.npy'), (), initial=saved_logl),
                          'saved_logwt': GrowingNpy(os.path.join(cp, 'saved_logwt.npy'), (), initial=saved_logwt)}
        self._grow['saved_v'].sync(saved_v)
        self._grow['saved_logl'].sync(saved_logl)
        self._grow['saved_logwt'].sync(saved_logwt)
        with open(os.path.join(cp, 'checkpoint_%s.txt' % it), 'w') as f:
            json.dump(state, f)

    def _native_prior_ok(self, strategy, expired_strategies, rejection_trials):
        """the 'rejection_prior' phase can run in the native loop too: block-evaluated candidates (the fused likelihood kernel), one
        process, no derived parameters, and 'mcmc' as the strategy that follows (the native loop runs on into it)"""
        rest = [m for m in strategy if m not in expired_strategies and m != 'rejection_prior']
        return (self._fused_like_id is not None and rejection_trials is None and not self.use_mpi and self.num_derived == 0
                and len(rest) > 0 and rest[0] == 'mcmc')

    def _mcmc_loop_native(self, loc, strategy, expired_strategies, mcmc_steps, C, dynamic, step_size, train_iters, jitter,
                          update_interval, log_interval, dlogz, max_iters, primary, prior_phase=False, volume_switch=-1.0):
        """The rest of run()'s while loop once 'mcmc' is the strategy in force: nnest_host_mcmc_consume does the per-iteration
        body (nested.py:269-293, :429-437, :458-471); this method does what it returns for -- retrain (nested.py:311-314), a new
        batch of chains (nested.py:399-427), the log line + results.csv row (nested.py:439-456), the checkpoint
        (nested.py:473-485).  `loc`: run()'s locals at the hand-over.  The trajectory is the Python loop's, value for value
        (tests/test_reference_trajectory.py runs both)."""
        lib = _lib.load()
        N, D, nd = self.num_live_points, self.x_dim, self.num_derived
        W = D + nd
        f64 = lambda a, shape: np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(shape))
        active_u, active_v = f64(loc['active_u'], (N, D)), f64(loc['active_v'], (N, D))
        active_logl, active_derived = f64(loc['active_logl'], (N,)), f64(loc['active_derived'], (N, nd))
        ev = loc['ev']
        n0 = len(loc['saved_logl'])
        cap = max(2 * n0 + 4096, 1 << 16)
        dead = {'v': np.empty((cap, W)), 'logl': np.empty(cap), 'logwt': np.empty(cap), 'zprev': np.empty(cap)}
        if n0:
            dead['v'][:n0] = np.asarray(loc['saved_v'], dtype=np.float64).reshape(n0, W)
            dead['logl'][:n0], dead['logwt'][:n0] = loc['saved_logl'], loc['saved_logwt']
        h_upto = n0                       # ev.h covers the dead points [0, h_upto)
        st = _lib.HostState(logz=ev.logz, logvol=loc['logvol'], fraction_remain=loc['fraction_remain'], max_logl=float(loc['max_logl']),
                            loglstar=0.0, it=int(loc['it']), n_dead=n0, accept_point=int(loc['accept_point']), nb=C,
                            first_time=int(loc['first_time']), resume=_lib.HOST_TOP, worst=0, pad_=0)
        end_u, end_v, end_logl = np.zeros((C, D)), np.zeros((C, D)), np.zeros(C)
        moved, end_derived = np.zeros(C, dtype=np.uint8), np.zeros((C, nd))
        total_calls, scale = loc['total_calls'], loc['scale']
        last_checkpoint = last_chain = time.time()
        if self.chain_min_seconds <= 0:
            last_chain = -1e300
        scalars_upto = st.it
        P = lambda a: a.ctypes.data_as(ctypes.c_void_p)

        def catch_up():
            """the information H (nested.py:281-283) over the dead points added since the last call -- the exponentials by numpy,
            as the reference forms them -- and log Z"""
            nonlocal h_upto
            n = st.n_dead
            if n > h_upto:
                zprev = dead['zprev'][h_upto:n]
                total = np.append(zprev[1:], st.logz)   # log Z after the update = log Z before the next one
                e1, e2 = np.exp(dead['logwt'][h_upto:n] - total), np.exp(zprev - total)
                logl = np.ascontiguousarray(dead['logl'][h_upto:n])
                ev.h = float(lib.nnest_host_h_update(ev.h, P(e1), P(e2), P(logl), P(np.ascontiguousarray(zprev)), P(total), n - h_upto))
                h_upto = n
            ev.logz = st.logz

        def scalars():
            """trainer.writer.add_scalar('logz', ev.logz, it) of every accepted point (nested.py:467), in bulk: at step k log Z
            holds k dead points"""
            nonlocal scalars_upto
            if primary and st.it > scalars_upto:
                ks = np.arange(scalars_upto + 1, st.it + 1)
                z = np.where(ks < st.n_dead, dead['zprev'][np.minimum(ks, st.n_dead - 1)], st.logz)
                if hasattr(self.trainer.writer, 'add_scalars'):
                    self.trainer.writer.add_scalars('logz', ks, z)
                else:
                    for k, v in zip(ks, z):
                        self.trainer.writer.add_scalar('logz', float(v), int(k))
            scalars_upto = st.it

        # (the arrays live for the whole loop -- the dead-point arrays until they are grown: their addresses are taken once, not per
        # call: 2 600 calls x 13 ctypes casts were 0.07 s of a config-2 run)
        fixed = [P(a) for a in (active_u, active_v, active_logl, active_derived, end_u, end_v, end_logl, moved, end_derived)]
        deadp = [P(dead[k]) for k in ('v', 'logl', 'logwt', 'zprev')]
        stp = ctypes.byref(st)
        results_f = None

        def grow_dead():
            nonlocal cap, deadp
            cap *= 2
            for k, a in dead.items():
                b = np.empty((cap,) + a.shape[1:])
                b[:a.shape[0]] = a
                dead[k] = b
            deadp = [P(dead[k]) for k in ('v', 'logl', 'logwt', 'zprev')]

        def checkpoint():                                    # nested.py:473-485
            nonlocal last_checkpoint, last_chain
            scalars()
            if primary and time.time() - last_checkpoint >= self.checkpoint_min_seconds:
                last_checkpoint = time.time()
                catch_up()
                n = st.n_dead
                self._checkpoint(st.it, active_u, active_v, active_logl, active_derived, dead['v'][:n], dead['logl'][:n],
                                 dead['logwt'][:n],
                                 {'logz': ev.logz, 'h': ev.h, 'logvol': st.logvol, 'ncall': total_calls,
                                  'fraction_remain': st.fraction_remain, 'strategy': strategy,
                                  'expired_strategies': expired_strategies})
                if last_checkpoint - last_chain >= self.chain_min_seconds:
                    last_chain = last_checkpoint
                    self.samples = np.array(dead['v'][:n])
                    self.weights = np.exp(dead['logwt'][:n] - ev.logz)
                    self.loglikes = np.array(dead['logl'][:n])
                    self._save_samples(self.samples, self.loglikes, weights=self.weights)

        try:
            finished = False
            if prior_phase:
                # 'rejection_prior' in force (nested.py:322-334, :362-373): nnest_host_prior_consume walks the candidates of a block of
                # prior draws evaluated by one launch of the likelihood kernel (the rule of Sampler._rejection_prior_sample above, value
                # for value) and returns for a new block, the log line, the checkpoint and when the strategy expires
                from . import flow
                pr = _lib.HostPrior()
                pr.block_next = int(getattr(self, '_prior_block', 256))
                pr.total_calls = int(self.total_calls)
                prp = ctypes.byref(pr)
                mcmc_valid = int('mcmc' in strategy and 'mcmc' not in expired_strategies)
                i64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)
                blk = dict(idx=i64(np.zeros(1)), l32=np.zeros(1), l64=np.zeros(1), u=np.zeros((1, D)), v=np.zeros((1, D)), d=np.zeros((1, max(nd, 1))))
                blkp = [P(blk[k]) for k in ('idx', 'l32', 'l64', 'u', 'v', 'd')]
                while True:
                    reason = lib.nnest_host_prior_consume(stp, prp, N, D, nd, *fixed[:4], *blkp, *deadp, cap, float(dlogz), int(max_iters),
                                                          int(log_interval), float(volume_switch), float(mcmc_steps), mcmc_valid)
                    self.total_calls = total_calls = int(pr.total_calls)
                    if pr.expired and 'rejection_prior' not in expired_strategies:   # nested.py:328-334 (said where the reference says it:
                        if primary:                                                  # before the pass's log line)
                            self.logger.info('Rejection prior no longer efficient, switching sampling method')
                        expired_strategies.append('rejection_prior')
                    if reason == _lib.HOST_FINISHED:
                        finished = True
                        break
                    if reason == _lib.HOST_EXPIRED:
                        break
                    if reason == _lib.HOST_DEAD_FULL:
                        grow_dead()
                        st.resume = _lib.HOST_TOP
                    elif reason == _lib.HOST_NEED_SAMPLES:       # a block of prior draws, one launch of the likelihood kernel
                        x = self.sample_prior(int(pr.block_next))
                        logl = flow.loglike(self._fused_like_id, x, self._linear_scale, device=self.trainer.netG.device,
                                            like_params=self._fused_like_params).cpu().numpy()
                        cand = np.flatnonzero(logl > st.loglstar)
                        if len(cand):   # the kernel works on float32(x); the stored value is the reference's float64 one
                            xc = np.ascontiguousarray(x[cand], dtype=np.float64)
                            calls = self.total_calls
                            l64, _d64 = self.loglike(xc)
                            self.total_calls = calls
                            blk = dict(idx=i64(cand), l32=np.ascontiguousarray(logl[cand], dtype=np.float64),
                                       l64=np.ascontiguousarray(l64, dtype=np.float64), u=xc,
                                       v=np.ascontiguousarray(self.transform(xc), dtype=np.float64), d=np.zeros((1, max(nd, 1))))
                            blkp = [P(blk[k]) for k in ('idx', 'l32', 'l64', 'u', 'v', 'd')]
                        pr.n, pr.n_cand, pr.pos, pr.k, pr.hits = len(logl), len(cand), 0, 0, 0
                        st.resume = _lib.HOST_AFTER_SAMPLES
                    elif reason == _lib.HOST_LOG:                # nested.py:374-378 (before `it` advances)
                        if primary:
                            self.logger.info('Step [%d] loglstar [%5.4e] max logl [%5.4e] logz [%5.4e] vol [%6.5e] ncalls [%d] '
                                             'mean calls [%5.4f]' % (st.it + 1, st.loglstar, st.max_logl, st.logz, np.exp(-st.it / N),
                                                                     total_calls, pr.mean_calls))
                        st.resume = _lib.HOST_AFTER_LOG
                    elif reason == _lib.HOST_CHECKPOINT:
                        checkpoint()
                        st.resume = _lib.HOST_TOP
                    else:
                        raise RuntimeError('nnest_host_prior_consume returned %d' % reason)
                self._prior_block = int(pr.block_next)
                st.nb, st.resume = C, _lib.HOST_TOP              # the MCMC strategy starts with a batch of its own (nested.py:307-309)
            while not finished:
                reason = lib.nnest_host_mcmc_consume(stp, N, D, nd, *fixed, C, *deadp, cap, float(dlogz),
                                                     int(max_iters), int(update_interval), int(log_interval))
                if reason == _lib.HOST_FINISHED:
                    break
                if reason == _lib.HOST_DEAD_FULL:
                    grow_dead()
                    st.resume = _lib.HOST_TOP
                elif reason == _lib.HOST_RETRAIN:
                    self._train(active_u, train_iters, jitter)   # nested.py:311-314
                    self.num_retrains += 1
                    st.first_time = 0
                    st.resume = _lib.HOST_AFTER_TRAIN
                elif reason == _lib.HOST_NEED_SAMPLES:           # nested.py:399-427
                    per = -(-C // self.mpi_size)
                    ctl = np.zeros(per * self.mpi_size + 1, dtype=np.int64)
                    if primary:
                        idx = np.random.randint(low=0, high=N, size=C)      # nested.py:405
                        ctl[:-1] = np.resize(idx, per * self.mpi_size)
                        if self._fused_like_id is not None:   # only the HIP kernel's noise streams consume a seed
                            ctl[-1] = self._next_seed() & 0x7FFFFFFFFFFFFFFF
                    ctl = self._broadcast(ctl)
                    lo = self.mpi_rank * per
                    my = ctl[lo:lo + per]
                    kw = dict(init_samples=active_u[my, :], init_loglikes=active_logl[my], loglstar=st.loglstar,
                              walker_offset=lo, seed=int(ctl[-1]), form=self._pinned_form(C, dynamic))
                    if self._fused_like_id is not None and nd == 0:
                        ends, scale, nc = self._mcmc_endpoints_fused(mcmc_steps, step_size, dynamic, **kw)
                    else:
                        s_x, _lat, s_d, s_l, scale, nc = self._mcmc_sample(
                            mcmc_steps, step_size=step_size, dynamic_step_size=dynamic,
                            init_derived=active_derived[my, :] if nd > 0 else np.empty((per, 0)), **kw)
                        mv = np.all(s_x[:, 0, :] != s_x[:, -1, :], axis=1)   # a chain is usable if every coordinate moved (nested.py:432)
                        ends = np.concatenate([s_x[:, -1, :], s_l[:, -1:], mv[:, None], s_d[:, -1, :]], axis=1).astype(np.float64)
                    ends = self._all_gather_rows(ends)[:C]
                    ends = ends.cpu().numpy() if torch.is_tensor(ends) else ends
                    end_u[:] = ends[:, :D]
                    end_logl[:] = ends[:, D]
                    moved[:] = ends[:, D + 1] != 0
                    if nd > 0:
                        end_derived[:] = ends[:, D + 2:]
                    end_v[:] = self.transform(end_u)
                    self.num_batches += 1
                    total_calls = int(self._all_sum(self.total_calls))
                    st.nb = 0
                    st.resume = _lib.HOST_AFTER_SAMPLES
                elif reason == _lib.HOST_LOG:                    # nested.py:439-456 (before `it` advances)
                    if primary:
                        acc = self.total_accepted / max(1, self.total_accepted + self.total_rejected)
                        self.logger.info('Step [%d] loglstar [%5.4e] maxlogl [%5.4e] logz [%5.4e] vol [%6.5e] ncalls [%d] '
                                         'scale [%5.4f]' % (st.it, st.loglstar, st.max_logl, st.logz, np.exp(-st.it / N), total_calls, scale))
                        # (one handle for the loop, flushed row by row: a config-2 run appends 1000 rows, and open/close per row was 25 ms)
                        if results_f is None:
                            results_f = open(os.path.join(self.logs['results'], 'results.csv'), 'a')
                        csv.writer(results_f).writerow([st.it, acc, float('nan'), float('nan'), float('nan'), scale, np.float64(st.loglstar),
                                                        np.float64(st.logz), np.float64(st.fraction_remain), total_calls])
                        results_f.flush()
                    st.resume = _lib.HOST_AFTER_LOG
                elif reason == _lib.HOST_CHECKPOINT:             # nested.py:473-485
                    checkpoint()
                    st.resume = _lib.HOST_TOP
                else:
                    raise RuntimeError('nnest_host_mcmc_consume returned %d' % reason)
        finally:   # (the results.csv handle of the loop, also when a callback raises)
            if results_f is not None:
                results_f.close()
        scalars()
        catch_up()
        n = st.n_dead
        # (the dead points as ARRAYS: 2e5 of them at config 2 -- as three Python lists they cost 0.1 s to build and to turn back)
        return (active_u, active_v, active_logl, active_derived, dead['v'][:n], dead['logl'][:n], dead['logwt'][:n],
                ev, st.logvol, st.fraction_remain, int(st.it), total_calls, scale)

    # ---- the run ----------------------------------------------------------------------------------------------
    def run(self,
            strategy=None,
            mcmc_steps=0,
            mcmc_num_chains=10,
            mcmc_dynamic_step_size=True,
            max_iters=1000000,
            update_interval=None,
            log_interval=None,
            dlogz=0.5,
            train_iters=500,
            volume_switch=-1.0,
            step_size=0.0,
            jitter=-1.0,
            rejection_cache_interval=10,
            rejection_enlargement_factor=1.1,
            rejection_trials=None,
            mcmc_step_lag=None,
            mcmc_step_warm=None):
        # mcmc_step_lag (not in the reference): steps between an MCMC step and the proposal scale that reflects its batch-wide
        # accept count inside the HIP kernel; 0 = the reference's rule exactly (one grid-wide wait per step), None = the
        # default that keeps the wait off the step (include/nnest_hip.h NNEST_MH_LAG)
        self.mcmc_step_lag = mcmc_step_lag
        # mcmc_step_warm (not in the reference): with a lagged rule, the first so many steps of every launch apply it exactly
        # (NNEST_MH_WARM); None = the kernel form's default
        self.mcmc_step_warm = mcmc_step_warm
        if hasattr(self.trainer, 'wait_for_saves'):
            self.trainer.async_save = True   # models/netG.pt written beside the GPU work; run() waits for the last one below
            if hasattr(self.trainer, 'background_jobs') and hasattr(self.trainer.writer, 'jobs'):
                self.trainer.writer.jobs = self.trainer.background_jobs()   # ... and the bulk log-Z scalars
        self._grow = None   # the dead-point files of a previous run() on this object are not this run's
        self._prior_cache = None   # (nor are the prior candidates it left unexamined: a reseeded run must not depend on them)
        if strategy is None or len(strategy) == 0:
            strategy = ['rejection_prior', 'mcmc']
        for s in strategy:
            if s not in ('rejection_prior', 'rejection_flow', 'density_flow', 'mcmc'):
                raise NotImplementedError("strategy %r: this build implements 'rejection_prior', 'rejection_flow', "
                                          "'density_flow' and 'mcmc'" % s)
        expired_strategies = []
        current_method = ''
        N = self.num_live_points
        update_interval = max(1, round(0.5 * N)) if update_interval is None else round(update_interval)
        if update_interval < 1:
            raise ValueError('update_interval must be >= 1')
        log_interval = max(1, round(0.2 * N)) if log_interval is None else round(log_interval)
        if log_interval < 1:
            raise ValueError('log_interval must be >= 1')
        if mcmc_steps <= 0:
            mcmc_steps = 5 * self.x_dim                      # nested.py:155-156
        if step_size <= 0.0:
            step_size = 1 / self.x_dim ** 0.5                # nested.py:158-159
        primary = self.single_or_primary_process
        if primary:
            self.logger.info('MCMC steps [%d]' % mcmc_steps)
            self.logger.info('Initial scale [%5.4f]' % step_size)
            self.logger.info('Volume switch [%5.4f]' % volume_switch)

        it = -1
        if self.resume and self.logs is not None and not self.logs['created']:
            for f in glob.glob(os.path.join(self.logs['checkpoint'], 'checkpoint_*.txt')):
                it = max(it, int(f.split('/checkpoint_')[1].split('.txt')[0]))
        if self.use_mpi:
            it = int(self._broadcast(np.array([it], dtype=np.int64))[0])
        total_calls = 0
        if it >= 0:
            # nested.py:166-195.  The reference has every MPI rank read the checkpoint files; here rank 0 reads them and
            # broadcasts the state (ranks need no shared file system), each rank taking ncall / size as its own call count
            # (nested.py:183)
            STRATS = ('rejection_prior', 'rejection_flow', 'density_flow', 'mcmc')
            nd = self.num_derived
            if primary:
                self.logger.info('Using checkpoint [%d]' % it)
                cp = self.logs['checkpoint']
                with open(os.path.join(cp, 'checkpoint_%s.txt' % it), 'r') as f:
                    data = json.load(f)
                active_u = np.load(os.path.join(cp, 'active_u_%s.npy' % it))
                active_logl = np.load(os.path.join(cp, 'active_logl_%s.npy' % it))
                active_derived = np.load(os.path.join(cp, 'active_derived_%s.npy' % it))
                saved_arr = np.load(os.path.join(cp, 'saved_v.npy')).reshape(-1, self.x_dim + nd)
                saved_logl = np.load(os.path.join(cp, 'saved_logl.npy'))
                saved_logwt = np.load(os.path.join(cp, 'saved_logwt.npy'))
                assert it == len(saved_logl)
                head = np.array([data['logz'], data['h'], data['logvol'], data['ncall'], data['fraction_remain'],
                                 len(data['strategy']), len(data['expired_strategies'])]
                                + [STRATS.index(m) for m in data['strategy']] + [-1] * (4 - len(data['strategy']))
                                + [STRATS.index(m) for m in data['expired_strategies']] + [-1] * (4 - len(data['expired_strategies'])),
                                dtype=np.float64)
            else:
                head = np.empty(15)
                active_u, active_logl = np.empty((N, self.x_dim)), np.empty(N)
                active_derived = np.empty((N, nd))
                saved_arr, saved_logl, saved_logwt = np.empty((it, self.x_dim + nd)), np.empty(it), np.empty(it)
            if self.use_mpi:
                head = self._broadcast(head)
                active_u, active_logl = self._broadcast(active_u), self._broadcast(active_logl)
                if nd > 0:
                    active_derived = self._broadcast(active_derived)
                if it > 0:
                    saved_arr, saved_logl, saved_logwt = (self._broadcast(np.ascontiguousarray(a, dtype=np.float64))
                                                          for a in (saved_arr, saved_logl, saved_logwt))
            data = dict(logz=float(head[0]), h=float(head[1]), logvol=float(head[2]), ncall=int(head[3]), fraction_remain=float(head[4]),
                        strategy=[STRATS[int(k)] for k in head[7:7 + int(head[5])]],
                        expired_strategies=[STRATS[int(k)] for k in head[11:11 + int(head[6])]])
            saved_v, saved_logl, saved_logwt = list(saved_arr), list(saved_logl), list(saved_logwt)
            ev, logvol = _Evidence(data['logz'], data['h']), data['logvol']
            self.total_calls = int(data['ncall'] / self.mpi_size)
            total_calls = data['ncall']
            fraction_remain = data['fraction_remain']
            strategy, expired_strategies = data['strategy'], data['expired_strategies']
            active_v = self.transform(active_u)
        else:
            active_u = self.sample_prior(N) if primary else np.empty((N, self.x_dim))   # nested.py:199-207
            active_u = self._broadcast(active_u)
            active_v = self.transform(active_u)
            active_logl, active_derived = self._initial_loglikes(active_u)
            total_calls = int(self._all_sum(self.total_calls))
            if primary:
                self.logger.info('Step [0] max logl [%5.4e] vol [1.0] ncalls [%d]' % (np.max(active_logl), total_calls))
            saved_v, saved_logl, saved_logwt = [], [], []
            ev = _Evidence()
            logvol = np.log(1.0 - np.exp(-1.0 / N))          # nested.py:244
            fraction_remain = 1.0
            it = 0
            if primary:
                self._checkpoint(it, active_u, active_v, active_logl, active_derived, saved_v, saved_logl, saved_logwt,
                                 {'logz': ev.logz, 'h': ev.h, 'logvol': logvol, 'ncall': total_calls,
                                  'fraction_remain': fraction_remain, 'strategy': strategy,
                                  'expired_strategies': expired_strategies})

        last_checkpoint = last_chain = time.time()
        if self.chain_min_seconds <= 0:
            last_chain = -1e300
        first_time = True
        get_samples = True
        nb = 0
        ncs = []
        accept_point = True
        samples = loglikes = None
        scale = step_size
        mean_calls = 0
        self.num_retrains = 0
        self.num_batches = 0

        # (the largest live likelihood, kept up to date: the point that is replaced is the smallest one, so the maximum only grows;
        # the reference takes np.max over the live points at every iteration, nested.py:461 -- the same value)
        max_logl = np.max(active_logl)
        while fraction_remain > dlogz and it <= max_iters:
            live = _first_live(strategy, expired_strategies)
            if self.native_loop and (live == 'mcmc' or (live == 'rejection_prior' and current_method in ('', 'rejection_prior')
                                                        and self._native_prior_ok(strategy, expired_strategies, rejection_trials))):
                # From here on the strategy in force is 'mcmc' for good (strategies only expire, nested.py:300-306, and 'mcmc'
                # never does): the per-iteration body runs in the native library, which returns for everything that is not
                # that arithmetic (nnest_host_mcmc_consume, include/nnest_hip.h)
                native = self._mcmc_loop_native(
                    locals(), strategy, expired_strategies, mcmc_steps, mcmc_num_chains, mcmc_dynamic_step_size, step_size, train_iters,
                    jitter, update_interval, log_interval, dlogz, max_iters, primary, prior_phase=live == 'rejection_prior',
                    volume_switch=volume_switch)
                (active_u, active_v, active_logl, active_derived, saved_v, saved_logl, saved_logwt, ev, logvol, fraction_remain, it,
                 total_calls, scale) = native
                break
            worst = int(np.argmin(active_logl))              # nested.py:272
            logwt = logvol + active_logl[worst]
            loglstar = active_logl[worst]
            if accept_point:
                # the worst live point dies: it joins the evidence and the chain (nested.py:280-293)
                ev.add(logwt, loglstar)
                dead = active_v[worst] if self.num_derived == 0 else np.concatenate((active_v[worst], active_derived[worst]))
                saved_v.append(np.array(dead, copy=True))
                saved_logwt.append(logwt)
                saved_logl.append(loglstar)
                accept_point = False

            method = _first_live(strategy, expired_strategies)
            if method != current_method:
                current_method, get_samples = method, True

            if current_method != 'rejection_prior' and (first_time or it % update_interval == 0):
                self._train(active_u, train_iters, jitter)   # nested.py:311-314
                self.num_retrains += 1
                first_time = False

            if current_method in ('rejection_prior', 'rejection_flow', 'density_flow'):   # nested.py:322-396
                if get_samples:
                    nb = 0
                    nd = self.num_derived
                    pack = None
                    if primary:
                        if current_method == 'rejection_prior':
                            s_x, s_l, s_d, nc = self._rejection_prior_sample(loglstar, num_trials=rejection_trials)
                        elif current_method == 'rejection_flow':     # nested.py:336-341
                            s_x, s_l, s_d, nc = self._rejection_flow_sample(
                                active_u, loglstar, enlargement_factor=rejection_enlargement_factor,
                                # as the reference: the envelope is REcomputed unless this flag is set (sampler.py:563-569)
                                cache=it % rejection_cache_interval == 0 or it % update_interval == 0)
                        else:                                        # nested.py:350-352
                            s_x, s_l, s_d, nc = self._density_sample(loglstar)
                        s_x, s_l = np.atleast_2d(s_x), np.ravel(s_l)
                        s_d = np.empty((len(s_l), 0)) if nd == 0 else np.asarray(s_d, dtype=np.float64).reshape(len(s_l), nd)
                    if self.use_mpi:
                        if primary:   # one row per candidate: [x | logl | derived], then the call count
                            pack = np.concatenate([np.concatenate([s_x, s_l[:, None], s_d], axis=1).ravel(), [float(nc)]])
                        n_rows = rejection_trials if (rejection_trials and current_method == 'rejection_prior') else 1
                        if not primary:
                            pack = np.empty(n_rows * (self.x_dim + 1 + nd) + 1)
                        pack = self._broadcast(pack)
                        rows = pack[:-1].reshape(-1, self.x_dim + 1 + nd)
                        samples, loglikes, derived_samples = rows[:, :self.x_dim], rows[:, self.x_dim], rows[:, self.x_dim + 1:]
                        nc = pack[-1]
                    else:         # (nothing to hand to other ranks: the candidates as they came -- 5 300 iterations of a config-2 run)
                        samples, loglikes, derived_samples, nc = s_x, s_l, s_d, float(nc)
                    ncs.append(nc)
                    mean_calls = np.mean(ncs[-20:]) if len(ncs) > 20 else 0
                    mcmc_valid = 'mcmc' in strategy and 'mcmc' not in expired_strategies
                    if current_method == 'rejection_prior':
                        expire = np.exp(-it / N) < volume_switch >= 0 or (volume_switch < 0 and mean_calls > mcmc_steps
                                                                       and mcmc_valid)
                    else:                                            # nested.py:344-347, :355-358
                        expire = mean_calls > mcmc_steps and mcmc_valid
                    if expire:
                        if primary:
                            self.logger.info('%s no longer efficient, switching sampling method'
                                             % current_method.replace('_', ' ').capitalize())
                        expired_strategies.append(current_method)
                        ncs = []
                while nb < samples.shape[0]:          # consume the candidates in order (nested.py:362-373)
                    cand = nb
                    nb += 1
                    get_samples = nb == samples.shape[0]
                    if loglikes[cand] > loglstar:
                        active_u[worst] = samples[cand]
                        active_v[worst] = self.transform(active_u[worst])
                        active_logl[worst] = loglikes[cand]
                        max_logl = max(max_logl, active_logl[worst])
                        if self.num_derived > 0:
                            active_derived[worst] = derived_samples[cand]
                        accept_point = True
                        break
                total_calls = int(self._all_sum(self.total_calls))
                if accept_point and it > 0 and (it + 1) % log_interval == 0 and primary:
                    self.logger.info('Step [%d] loglstar [%5.4e] max logl [%5.4e] logz [%5.4e] vol [%6.5e] ncalls [%d] '
                                     'mean calls [%5.4f]' % (it + 1, loglstar, max_logl, ev.logz, np.exp(-it / N),
                                                             total_calls, mean_calls))

            elif current_method == 'mcmc':                   # nested.py:398-456
                if get_samples:
                    nb = 0
                    C, D, nd = mcmc_num_chains, self.x_dim, self.num_derived
                    per = -(-C // self.mpi_size)
                    ctl = np.zeros(per * self.mpi_size + 1, dtype=np.int64)
                    if primary:
                        idx = np.random.randint(low=0, high=N, size=C)      # nested.py:405
                        ctl[:-1] = np.resize(idx, per * self.mpi_size)
                        if self._fused_like_id is not None:   # only the HIP kernel's noise streams consume a seed
                            ctl[-1] = self._next_seed() & 0x7FFFFFFFFFFFFFFF
                    ctl = self._broadcast(ctl)
                    lo = self.mpi_rank * per
                    my = ctl[lo:lo + per]
                    kw = dict(init_samples=active_u[my, :], init_loglikes=active_logl[my], loglstar=loglstar,
                              walker_offset=lo, seed=int(ctl[-1]), form=self._pinned_form(C, mcmc_dynamic_step_size))
                    # what the loop below consumes of a chain is its last x, the last logL and whether every coordinate moved
                    # (+ derived): one row [x_S | logL_S | moved | derived_S] per chain, all-gathered over the ranks (C2) -- on
                    # device memory when the whole batch ran inside the HIP kernel
                    if self._fused_like_id is not None and nd == 0:
                        ends, scale, nc = self._mcmc_endpoints_fused(mcmc_steps, step_size, mcmc_dynamic_step_size, **kw)
                    else:
                        s_x, _lat, s_d, s_l, scale, nc = self._mcmc_sample(
                            mcmc_steps, step_size=step_size, dynamic_step_size=mcmc_dynamic_step_size,
                            init_derived=active_derived[my, :] if nd > 0 else np.empty((per, 0)), **kw)
                        mv = np.all(s_x[:, 0, :] != s_x[:, -1, :], axis=1)   # a chain is usable if every coordinate moved (nested.py:432)
                        ends = np.concatenate([s_x[:, -1, :], s_l[:, -1:], mv[:, None], s_d[:, -1, :]], axis=1).astype(np.float64)
                    ends = self._all_gather_rows(ends)[:C]
                    ends = ends.cpu().numpy() if torch.is_tensor(ends) else ends
                    end_u, end_logl, moved, end_derived = ends[:, :D], ends[:, D], ends[:, D + 1] != 0, ends[:, D + 2:]
                    self.num_batches += 1
                    # (its transform once per batch)
                    end_v = self.transform(end_u)
                while nb < C:
                    cand = nb
                    nb += 1
                    get_samples = nb == C
                    if moved[cand] and end_logl[cand] > loglstar:    # nested.py:432-437
                        active_u[worst] = end_u[cand]
                        active_v[worst] = end_v[cand]
                        active_logl[worst] = end_logl[cand]
                        max_logl = max(max_logl, active_logl[worst])
                        if nd > 0:
                            active_derived[worst] = end_derived[cand]
                        accept_point = True
                        break
                total_calls = int(self._all_sum(self.total_calls))
                if accept_point and it > 0 and it % log_interval == 0 and primary:
                    acc = self.total_accepted / max(1, self.total_accepted + self.total_rejected)
                    self.logger.info('Step [%d] loglstar [%5.4e] maxlogl [%5.4e] logz [%5.4e] vol [%6.5e] ncalls [%d] '
                                     'scale [%5.4f]' % (it, loglstar, max_logl, ev.logz, np.exp(-it / N), total_calls, scale))
                    with open(os.path.join(self.logs['results'], 'results.csv'), 'a') as f:
                        csv.writer(f).writerow([it, acc, float('nan'), float('nan'), float('nan'), scale, loglstar, ev.logz,
                                                fraction_remain, total_calls])

            if accept_point:                                 # nested.py:458-485
                logvol -= 1.0 / N
                logz_remain = max_logl - it / N
                fraction_remain = np.logaddexp(ev.logz, logz_remain) - ev.logz
                it += 1
                if primary:
                    self.trainer.writer.add_scalar('logz', ev.logz, it)
                # checkpoint cadence as the reference (every log_interval accepted points, nested.py:473-485), but at
                # most one full dump per `checkpoint_min_seconds` (module docstring)
                if (it > 0 and it % log_interval == 0 and primary
                        and time.time() - last_checkpoint >= self.checkpoint_min_seconds):
                    last_checkpoint = time.time()
                    self._checkpoint(it, active_u, active_v, active_logl, active_derived, saved_v, saved_logl, saved_logwt,
                                     {'logz': ev.logz, 'h': ev.h, 'logvol': logvol, 'ncall': total_calls,
                                      'fraction_remain': fraction_remain, 'strategy': strategy,
                                      'expired_strategies': expired_strategies})
                    if last_checkpoint - last_chain >= self.chain_min_seconds:
                        last_chain = last_checkpoint
                        self.samples = np.array(saved_v)
                        self.weights = np.exp(np.array(saved_logwt) - ev.logz)
                        self.loglikes = np.array(saved_logl)
                        self._save_samples(self.samples, self.loglikes, weights=self.weights)

        # the remaining live points share the last volume shell equally (nested.py:487-495)
        logvol = -len(saved_v) / N - np.log(N)
        for i in range(N):
            ev.add(logvol + active_logl[i], active_logl[i])
        n_dead, W = len(saved_logl), self.x_dim + self.num_derived
        last_v = np.asarray(active_v, dtype=np.float64) if self.num_derived == 0 else np.concatenate((active_v, active_derived), axis=1)
        saved_v = np.concatenate((np.asarray(saved_v, dtype=np.float64).reshape(n_dead, W), last_v.reshape(N, W)))
        saved_logwt = np.concatenate((np.asarray(saved_logwt, dtype=np.float64).reshape(n_dead), logvol + np.asarray(active_logl, dtype=np.float64)))
        saved_logl = np.concatenate((np.asarray(saved_logl, dtype=np.float64).reshape(n_dead), np.asarray(active_logl, dtype=np.float64)))

        if hasattr(self.trainer, 'wait_for_saves'):
            self.trainer.wait_for_saves()
            self.trainer.async_save = False
            if getattr(self.trainer.writer, 'jobs', None) is not None:
                self.trainer.writer.jobs = None
        logz, h = ev.logz, ev.h
        self.logz = logz
        self.h = h
        self.logzerr = np.sqrt(h / N)
        self.niter = it + 1
        self.ncall = total_calls
        self.samples = np.array(saved_v)
        self.weights = np.exp(np.array(saved_logwt) - logz)
        self.loglikes = np.array(saved_logl)
        if primary:
            with open(os.path.join(self.logs['results'], 'final.csv'), 'w') as f:
                wr = csv.writer(f)
                wr.writerow(['niter', 'ncall', 'logz', 'logzerr', 'h'])
                wr.writerow([it + 1, total_calls, logz, np.sqrt(h / N), h])
            self._save_samples(self.samples, self.loglikes, weights=self.weights)
            self.trainer.writer.flush() if hasattr(self.trainer.writer, 'flush') else None
            self.logger.info('niter: {:d}\n ncall: {:d}\n nsamples: {:d}\n logz: {:6.3f} +/- {:6.3f}\n h: {:6.3f}'
                             .format(it + 1, total_calls, len(saved_v), logz, np.sqrt(h / N), h))
