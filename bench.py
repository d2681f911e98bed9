#!/usr/bin/env python3
"""bench.py -- live-point evals/s of the nnest hot path on MI355X.

One "step" = one launch of the persistent constrained-Metropolis kernel (K4, the GPU form of the
reference's Sampler._mcmc_sample, nnest/sampler.py:229-463) over the walker population of this rank:
  walkers x mcmc_steps proposals, each = one coupling-stack inverse (+log-det) of a 50-vector + box prior
  + one Rosenbrock log-likelihood  (= one "eval").
Workload = BASELINE.json configs[1]: Rosenbrock x_dim=50, 1000 live points (one walker per live point),
mcmc_steps = 5*x_dim = 250 (nnest/nested.py:155-156), NVP hidden 16 / 3 blocks / 1 layer.
Inputs are resident in HBM before the timed region.  Synthetic data: u ~ U(-1,1), seeded default-init
weights.  N>1: one process per GPU (torchrun), walkers sharded by rank with disjoint Philox streams,
no data-path collective (weak scaling: 1000 walkers per GPU).

  python bench.py --gpus 1 --steps 20 --warmup 3
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FP32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32-input MFMA = f32 vector peak
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E spec
MEASURED_TRAFFIC_BYTES = int((452.2 + 1384.7) * 1024)  # profiles/r01d/bench_pmc_summary.txt, per K4 launch


def useful_flops_per_eval(D, H, B, L):
    """SURVEY.md 8 table, mask-pruned count: B * 2 nets * 2 * (D/2*H + L*H^2 + H*D/2)"""
    return B * 4 * (D * H + L * H * H)


def alg_bytes_per_eval(D):
    """read z row + write x row + logdet + logl = 8D + 8"""
    return 8 * D + 8


def cpu_baseline(D, H, B, L, w, walkers, target_seconds=12.0):
    """The oracle (plain-C restatement of Sampler._mcmc_sample, single thread) timed on this box's host
    cores on a bounded sample of the same workload."""
    from oracle import oracle as orc
    o = orc.NVP(D, H, B, L, w)
    rng = np.random.RandomState(0)
    init = rng.uniform(-1, 1, size=(walkers, D))
    init_logl = orc.loglike('rosenbrock', init, 5.0)
    steps = 2
    dz = rng.normal(size=(steps, walkers, D)).astype(np.float32)
    u = rng.uniform(size=(steps, walkers)).astype(np.float32)
    t0 = time.perf_counter()
    orc.mcmc_sample(o, 'rosenbrock', 5.0, init, init_logl, float(init_logl.min()), 1 / np.sqrt(D), False, dz, u)
    dt = time.perf_counter() - t0
    per_step = dt / steps
    steps = int(max(2, min(5000, target_seconds / max(per_step, 1e-9))))
    dz = rng.normal(size=(steps, walkers, D)).astype(np.float32)
    u = rng.uniform(size=(steps, walkers)).astype(np.float32)
    t0 = time.perf_counter()
    orc.mcmc_sample(o, 'rosenbrock', 5.0, init, init_logl, float(init_logl.min()), 1 / np.sqrt(D), False, dz, u)
    dt = time.perf_counter() - t0
    return {'value': walkers * steps / dt, 'unit': 'evals/s', 'cores': 1, 'kind': 'port',
            'sample': '%d walkers x %d MH steps of the same workload, oracle/nnest_oracle.c single thread '
                      '(host has %d cores), %.1f s' % (walkers, steps, os.cpu_count(), dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--x-dim', type=int, default=50)
    ap.add_argument('--walkers', type=int, default=1000, help='walkers (live points) per GPU')
    ap.add_argument('--mcmc-steps', type=int, default=0, help='MH steps per launch (0 = 5*x_dim)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-saturation', action='store_true')
    ap.add_argument('--no-spline', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit('bench.py needs an MI355X (no GPU visible); there is no CPU fallback')
    # one rank per GPU; NNEST_BENCH_BACKEND=gloo is a single-GPU smoke test of the multi-rank code path only
    backend = os.environ.get('NNEST_BENCH_BACKEND', 'nccl')
    dev_index = local_rank if backend == 'nccl' else local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    from nnest_amd import flow, _lib

    D, H, B, L = args.x_dim, 16, 3, 1
    C = args.walkers
    S = args.mcmc_steps if args.mcmc_steps > 0 else 5 * D
    nvp = flow.HipNVP(D, H, B, L, device=dev, seed=0)
    rng = np.random.RandomState(1234 + rank)
    u0 = rng.uniform(-1, 1, size=(C, D))
    z0, _ = nvp.forward(u0)
    logl0 = flow.loglike(0, u0, 5.0, device=dev)
    loglstar = float(logl0.min())
    step_size = 1.0 / np.sqrt(D)

    # state buffers are re-seeded outside the timed launches (clone is not part of the hot path)
    zs = [z0.clone() for _ in range(args.steps + args.warmup)]
    ls = [logl0.clone() for _ in range(args.steps + args.warmup)]

    def launch(i):
        return nvp.mh_steps(0, 5.0, zs[i], ls[i], loglstar, step_size, S, dynamic=False, seed=42 + i,
                            walker_offset=rank * C)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        launch(i)
    barrier()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev0[k].record()
        launch(args.warmup + k)
        ev1[k].record()
    barrier()
    dt = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    evals_per_launch = C * S
    total_evals = evals_per_launch * args.steps * world
    value = total_evals / dt

    out = None
    if rank == 0:
        fl = useful_flops_per_eval(D, H, B, L)
        achieved_tflops = evals_per_launch * fl / (kern_ms * 1e-3) / 1e12
        info = _lib.device_info()
        out = {
            'metric': 'live-point evals/sec (flow-transform + loglike), x_dim=%d' % D,
            'value': value, 'unit': 'evals/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'Rosenbrock x_dim=%d, %d live points (walkers) per GPU, %d MH steps per launch, '
                                   'NVP hidden=%d blocks=%d layers=%d' % (D, C, S, H, B, L),
                       'walkers_per_gpu': C, 'mcmc_steps': S, 'evals_per_step': evals_per_launch,
                       'parallelism': 'walker-sharded x%d, no data-path collective' % world},
            'roofline': {'bound': 'mfma', 'achieved': achieved_tflops, 'peak': FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved_tflops / FP32_PEAK_TFLOPS,
                         # HBM bytes per launch from the committed rocprofv3 PMC passes of this command
                         # (profiles/r01d/bench_pmc_summary.txt: FETCH_SIZE 452.2 KiB + WRITE_SIZE 1384.7 KiB, raw); only
                         # valid for the default workload, null otherwise
                         'traffic': MEASURED_TRAFFIC_BYTES if (D, C, S) == (50, 1000, 250) else None,
                         'kernel': 'mh_kernel_team' if (C + 15) // 16 <= info['num_cu'] else 'mh_kernel',
                         'kernel_ms': kern_ms, 'flops_per_eval': fl,
                         'hbm_frac_if_streamed': evals_per_launch * alg_bytes_per_eval(D) / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         'note': 'f32-input MFMA peak (= f32 vector peak); %d walker tiles on %d CUs: occupancy-limited '
                                 'at this population' % ((C + 15) // 16, info['num_cu'])},
            'device': info['name'],
        }
        if not args.no_saturation and world == 1:
            # the same kernel at a population that fills the chip (not the headline: BASELINE's config is 1000)
            Cs = 16 * 4 * info['num_cu'] * 8  # 8 walker tiles per SIMD
            us = np.random.RandomState(5).uniform(-1, 1, size=(Cs, D))
            zz, _ = nvp.forward(us)
            ll = flow.loglike(0, us, 5.0, device=dev)
            Ss = 25
            nvp.mh_steps(0, 5.0, zz, ll, float(ll.min()), step_size, Ss, seed=1)
            torch.cuda.synchronize(dev)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            nvp.mh_steps(0, 5.0, zz, ll, float(ll.min()), step_size, Ss, seed=2)
            b.record()
            torch.cuda.synchronize(dev)
            ms = a.elapsed_time(b)
            out['saturated'] = {'walkers': Cs, 'mcmc_steps': Ss, 'kernel_ms': ms, 'evals_per_s': Cs * Ss / (ms * 1e-3),
                                'tflops': Cs * Ss * fl / (ms * 1e-3) / 1e12,
                                'frac_of_fp32_peak': Cs * Ss * fl / (ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS}
        if not args.no_spline and world == 1:
            # the same workload on the reference's default flow (neural spline flow): reported beside,
            # never as `value` (BASELINE's metric is quoted on the RealNVP path)
            from nnest_amd.spline import HipSpline
            sp = HipSpline(D, H, B, seed=0)
            sp.actnorm_init(u0[:min(C, 100)])
            zsp, _ = sp.forward(u0)
            t_ms = []
            for k in range(3):
                zz, ll = zsp.clone(), logl0.clone()
                torch.cuda.synchronize(dev)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                sp.mh_steps(0, 5.0, zz, ll, loglstar, step_size, S, seed=7 + k, walker_offset=rank * C)
                e1.record()
                torch.cuda.synchronize(dev)
                t_ms.append(e0.elapsed_time(e1))
            ms = float(np.median(t_ms[1:]))
            out['spline_flow'] = {'kernel': 'spline_mh_kernel_team' if (C + 15) // 16 <= 2 * info['num_cu'] else 'spline_mh_kernel',
                                  'kernel_ms': ms, 'evals_per_s': C * S / (ms * 1e-3),
                                  'note': 'SingleSpeedSpline hidden=%d blocks=%d bins=8; VALU-bound on the spline arithmetic' % (H, B)}
            if C >= 200:  # its training epoch at this population (90 % train / 10 % validation, batch 100: trainer.py:159-176)
                nv = C // 10
                E = 20
                perms = torch.stack([torch.randperm(C - nv) for _ in range(E)]).int()
                kw = dict(seed=1, jitter=0.01, batch=100, patience=50)
                sp.train_epochs(u0[nv:], u0[:nv], perms[:2], None, max_epochs=2, **kw)   # allocations
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                res = sp.train_epochs(u0[nv:], u0[:nv], perms, None, max_epochs=E, **kw)
                torch.cuda.synchronize(dev)
                out['spline_flow']['train_ms_per_epoch'] = (time.perf_counter() - t0) / max(1, res['epochs_run']) * 1e3
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(D, H, B, L, nvp.store_packed(), C)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
