#!/usr/bin/env python3
"""bench.py -- live-point evals/s (+ log-Z error) of the nnest hot path on MI355X.

One "step" = one launch of the persistent constrained-Metropolis kernel (K4, the GPU form of the reference's
Sampler._mcmc_sample, nnest/sampler.py:229-463) over the walker population of this rank, with the reference's default
step-size adaptation (mcmc_dynamic_step_size=True, nnest/nested.py:102) applied over the whole batch as the product does:
  walkers x mcmc_steps proposals, each = one coupling-stack inverse (+log-det) of a D-vector + box prior + one
  log-likelihood  (= one "eval").
Workload (default) = BASELINE.json configs[1]: Rosenbrock x_dim=50, 1000 live points (one walker per live point),
mcmc_steps = 5*x_dim = 250 (nnest/nested.py:155-156), NVP hidden 16 / 3 blocks / 1 layer.  Inputs are resident in HBM
before the timed region.  Synthetic data: u ~ U(-1,1), seeded default-init weights.

N > 1 (one process per GPU, torchrun): the real per-batch data path of a sharded run -- K4 on this rank's shard of the
walkers, then ONE RCCL all-gather of the chains' ends [C/N, D+2] float64 (C2) -- inside the timed
region.  --scaling weak (default): 1000 walkers per GPU; --scaling strong: the configuration's population split over
the ranks (--config 4: Himmelblau x_dim=32, 4000 live points; --config 5: Rosenbrock x_dim=100, 8000 live points); as in
the product, the kernel form of the whole batch is pinned on the shards only under a fixed step (where it makes the
sharded batch the unsharded one bit for bit) -- under the step rule every rank runs the fastest form of its shard.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: bench.py LAUNCHES the N ranks itself (one fresh child process per
GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, before anything in the parent touches a GPU; the reference's own
multi-process entry is `mpirun` over nnest/sampler.py:165-177), relays rank 0's one JSON line and exits non-zero if a rank
failed.  Under an outer `torchrun` (WORLD_SIZE set) it is one of the ranks, as before.  The line carries n_gpus == rccl_ranks == N,
`collective_backend`, the weak value and `strong_config2`.

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
METRIC = 'live-point evals/sec (flow-transform + loglike) + log-Z error, x_dim=%d'   # BASELINE.json "metric"
CONFIGS = {2: ('rosenbrock', 50, 5.0, 1000), 3: ('gaussmix', 20, 10.0, 2000), 4: ('himmelblau', 32, 5.0, 4000),
           5: ('rosenbrock', 100, 5.0, 8000)}
LIKE_ID = {'rosenbrock': 0, 'gaussmix': 1, 'himmelblau': 2}


def useful_flops_per_eval(D, H, B, L):
    """SURVEY.md 8 table, mask-pruned count: B * 2 nets * 2 * (D/2*H + L*H^2 + H*D/2)"""
    return B * 4 * (D * H + L * H * H)


def alg_bytes_per_eval(D):
    """read z row + write x row + logdet + logl = 8D + 8"""
    return 8 * D + 8


def _cpu_mcmc(args):
    from oracle import oracle as orc
    w, D, like, scale, init, init_logl, loglstar, step, dz, u = args
    o = orc.NVP(D, 16, 3, 1, w)
    orc.mcmc_sample(o, like, scale, init, init_logl, loglstar, step, True, dz, u)
    return init.shape[0] * dz.shape[0]


def cpu_baseline(D, w, like, scale, walkers, target_seconds=10.0):
    """The oracle (plain-C restatement of Sampler._mcmc_sample) timed on this box's host cores on a bounded sample of the
    same workload: once on ONE thread, once with the walkers split over all cores (walkers are independent; the C code
    releases the GIL under ctypes)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    rng = np.random.RandomState(0)
    init = rng.uniform(-1, 1, size=(walkers, D))
    init_logl = orc.loglike(like, init, scale)
    loglstar, step = float(init_logl.min()), 1 / np.sqrt(D)

    def run(steps, threads):
        dz = rng.normal(size=(steps, walkers, D)).astype(np.float32)
        u = rng.uniform(size=(steps, walkers)).astype(np.float32)
        cuts = np.linspace(0, walkers, threads + 1).astype(int)
        jobs = [(w, D, like, scale, init[a:b], init_logl[a:b], loglstar, step, np.ascontiguousarray(dz[:, a:b]),
                 np.ascontiguousarray(u[:, a:b])) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
        t0 = time.perf_counter()
        if threads == 1:
            _cpu_mcmc(jobs[0])
        else:
            with ThreadPoolExecutor(threads) as ex:
                list(ex.map(_cpu_mcmc, jobs))
        return walkers * steps / (time.perf_counter() - t0)

    cores = os.cpu_count() or 1
    try:   # what this process may actually use: the affinity mask and the container's CPU quota (cgroup v2 cpu.max: "quota period")
        cores = min(cores, len(os.sched_getaffinity(0)))
        with open('/sys/fs/cgroup/cpu.max') as f:
            q, per = f.read().split()[:2]
        if q != 'max':
            cores = max(1, min(cores, int(np.ceil(float(q) / float(per)))))
    except (OSError, ValueError, AttributeError):
        pass
    r1 = run(2, 1)
    one = run(int(max(2, min(2000, target_seconds * r1 / walkers))), 1)
    threads = min(cores, walkers)
    ra = run(2, threads)
    steps_all = int(max(2, min(4000, target_seconds * ra / walkers)))
    allc = run(steps_all, threads)
    return {'value': allc, 'unit': 'evals/s', 'cores': threads, 'kind': 'port',
            'one_thread': one,
            'sample': '%d walkers x %d MH steps of the same workload (dynamic step rule), oracle/nnest_oracle.c, walkers '
                      'split over %d threads (the %d CPUs this container may use: affinity mask and cgroup quota); one_thread = the same code on one core' % (
                          walkers, steps_all, threads, cores)}


def committed_traffic(kernel, tag='r06'):
    """HBM bytes per K4 launch from this round's committed rocprofv3 PMC passes of THIS command (scripts/profile_bench.sh ->
    profiles/<tag>/bench_pmc.json); None when the profile is absent or was taken on another kernel than the one that just ran"""
    path = os.path.join(ROOT, 'profiles', tag, 'bench_pmc.json')
    if not os.path.exists(path):
        return None, None
    with open(path) as f:
        d = json.load(f)
    if d.get('kernel') != kernel:
        return None, None
    return d.get('traffic_bytes_per_launch'), os.path.relpath(path, ROOT)


def logz_report(dev, live_run):
    """The log-Z half of the metric at config 2: CPU path (oracle-backed host driver, tests/golden/logz_cpu_cfg2.json) against
    the GPU path on the same seeds (tests/golden/logz_gpu_cfg2.json, tools/run_logz_gpu.py), plus one live run."""
    out = {}
    cpu_p = os.path.join(ROOT, 'tests', 'golden', 'logz_cpu_cfg2.json')
    gpu_p = os.path.join(ROOT, 'tests', 'golden', 'logz_gpu_cfg2.json')
    if os.path.exists(cpu_p) and os.path.exists(gpu_p):
        with open(cpu_p) as f:
            c = json.load(f)
        with open(gpu_p) as f:
            g = json.load(f)
        # (runs of the two paths with the same seed are not correlated -- the per-seed differences scatter like independent runs --
        # so each mean is taken over every seed its fixture holds)
        cv, gv = np.array(c['logz']), np.array(g['logz'])
        se = lambda v: float(v.std(ddof=1) / np.sqrt(len(v))) if len(v) > 1 else None
        out.update(cpu_mean=float(cv.mean()), cpu_stderr=se(cv), cpu_seeds=len(cv), gpu_mean=float(gv.mean()), gpu_stderr=se(gv),
                   gpu_seeds=len(gv))
        out['delta'] = out['gpu_mean'] - out['cpu_mean']
        out['combined_stderr'] = float(np.hypot(out['gpu_stderr'] or 0.0, out['cpu_stderr'] or 0.0))
        # "resolved": the ensembles can tell a difference of 0.2 from 0 at three standard errors, and they agree within 0.1
        out['resolved'] = bool(out['combined_stderr'] <= 0.07 and abs(out['delta']) <= 0.1)
        out['step_rule'] = g.get('step_rule')
        lag0_p = os.path.join(ROOT, 'tests', 'golden', 'logz_gpu_cfg2_lag0.json')
        if os.path.exists(lag0_p):   # the reference's exact step rule (lag 0) on the GPU path
            with open(lag0_p) as f:
                g0 = np.array(json.load(f)['logz'])
            out['gpu_lag0'] = {'mean': float(g0.mean()), 'stderr': se(g0), 'seeds': len(g0), 'delta_vs_cpu': float(g0.mean() - out['cpu_mean']),
                               'combined_stderr': float(np.hypot(se(g0) or 0.0, out['cpu_stderr'] or 0.0))}
    # config 3 (GaussianMix x_dim 20, 2000 live points): one run scatters by 0.11 only, so there the +-0.1 criterion is resolved
    c3, g3 = os.path.join(ROOT, 'tests', 'golden', 'logz_cpu_cfg3.json'), os.path.join(ROOT, 'tests', 'golden', 'logz_gpu_cfg3.json')
    if os.path.exists(c3) and os.path.exists(g3):
        with open(c3) as f:
            c = json.load(f)
        with open(g3) as f:
            g = json.load(f)
        cz, gz = dict(zip(c['seeds'], c['logz'])), dict(zip(g['seeds'], g['logz']))
        seeds = sorted(set(cz) & set(gz))
        cv, gv = np.array([cz[k] for k in seeds]), np.array([gz[k] for k in seeds])
        if len(seeds) > 1:
            out['config3'] = {'cpu_mean': float(cv.mean()), 'gpu_mean': float(gv.mean()), 'delta': float(gv.mean() - cv.mean()),
                              'combined_stderr': float(np.hypot(cv.std(ddof=1), gv.std(ddof=1)) / np.sqrt(len(seeds))), 'n_seeds': len(seeds),
                              'analytic': -20.0 * float(np.log(20.0))}
    # config 1 at the reference's default mcmc_num_chains = 10 (nested.py:185) instead of one chain per live point (fixtures *_cfg11)
    c11, g11 = os.path.join(ROOT, 'tests', 'golden', 'logz_cpu_cfg11.json'), os.path.join(ROOT, 'tests', 'golden', 'logz_gpu_cfg11.json')
    if os.path.exists(c11) and os.path.exists(g11):
        with open(c11) as f:
            cv = np.array(json.load(f)['logz'])
        with open(g11) as f:
            gv = np.array(json.load(f)['logz'])
        out['config1_ten_chains'] = {'cpu_mean': float(cv.mean()), 'gpu_mean': float(gv.mean()), 'delta': float(gv.mean() - cv.mean()),
                                     'combined_stderr': float(np.hypot(cv.std(ddof=1) / np.sqrt(len(cv)), gv.std(ddof=1) / np.sqrt(len(gv)))),
                                     'cpu_seeds': len(cv), 'gpu_seeds': len(gv)}
    if live_run:
        import tempfile
        from nnest_amd.likelihoods import Rosenbrock
        from nnest_amd.nested import NestedSampler
        np.random.seed(0)
        torch.manual_seed(0)
        s = NestedSampler(50, Rosenbrock(50), transform=lambda x: 5.0 * x, log_dir=tempfile.mkdtemp(dir='/tmp'),
                          num_live_points=1000, log_level=40, flow='nvp')
        # where the wall time of the run goes: both wrapped calls end in a read-back (the training result / the batch's counts), so
        # the wall time around them is the kernel's time plus its launch and read-back
        from nnest_amd import flow as _flow
        split = {'k5_s': 0.0, 'k4_s': 0.0, 'k5_calls': 0, 'k4_calls': 0, 'epochs': 0}
        _te, _ef = _flow.HipNVP.train_epochs, type(s)._mcmc_endpoints_fused

        def te(self, *a, **k):
            t = time.perf_counter()
            r = _te(self, *a, **k)
            torch.cuda.synchronize()
            split['k5_s'] += time.perf_counter() - t
            split['k5_calls'] += 1
            split['epochs'] += int(r['epochs_run']) - int(k.get('epoch_offset', 0))
            return r

        def ef(self, *a, **k):
            t = time.perf_counter()
            r = _ef(self, *a, **k)
            split['k4_s'] += time.perf_counter() - t
            split['k4_calls'] += 1
            return r
        _flow.HipNVP.train_epochs, type(s)._mcmc_endpoints_fused = te, ef
        t0 = time.time()
        try:
            s.run(mcmc_num_chains=1000)
        finally:
            _flow.HipNVP.train_epochs, type(s)._mcmc_endpoints_fused = _te, _ef
        wall = time.time() - t0
        out['live_run'] = {'logz': float(s.logz), 'logzerr': float(s.logzerr), 'wall_s': wall, 'ncall': int(s.ncall),
                           'seed': 0, 'delta_vs_cpu_mean': (float(s.logz) - out['cpu_mean']) if 'cpu_mean' in out else None}
        out['e2e'] = {'what': 'BASELINE config 2 end to end (NestedSampler.run, 1000 walkers per batch, seed 0), wall seconds',
                      'wall_s': wall, 'k5_s': split['k5_s'], 'k4_s': split['k4_s'], 'host_s': wall - split['k5_s'] - split['k4_s'],
                      'k5_calls': split['k5_calls'], 'k5_epochs': split['epochs'], 'k4_launches': split['k4_calls'],
                      'niter': int(s.niter), 'retrains': int(s.num_retrains)}
    out['note'] = ('independent noise streams: one run scatters by logzerr ~ sqrt(H/N) ~ 0.43 around the ensemble mean; the '
                   '+-0.1 statement is on the means (tests/test_gpu_nested.py::test_committed_logz_fixtures_resolve_the_acceptance)')
    return out


def launch_ranks(n, argv, timeout_s=1800.0):
    """`--gpus n` without an outer launcher: n fresh child processes, one per GPU (never an exec of a process that has touched
    a GPU: the parent has not), rank 0's stdout relayed.  ALL children are polled: the first one that exits non-zero is named, its
    siblings are terminated (they would otherwise sit in the rendezvous or in a collective until the store's timeout, minutes) and
    the parent exits non-zero within seconds; `--launch-timeout` bounds the whole run the same way.  Each child gets its own
    process group, so a teardown takes the rank's helpers with it and nothing else."""
    import socket
    import subprocess
    import tempfile
    with socket.socket() as sk:   # a free rendezvous port on the loopback (probed, then closed: taken again by rank 0's store)
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs, out0 = [], tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=os.environ.get('MASTER_PORT', str(port)), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, start_new_session=True,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))

    def teardown():
        import signal
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)
                except (ProcessLookupError, PermissionError):
                    pass
        t_end = time.monotonic() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except (ProcessLookupError, PermissionError):
                    pass
                p.wait()

    t0, failed = time.monotonic(), None
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed = 'rank %d exited with code %d' % bad[0] + (' (and %d more)' % (len(bad) - 1) if len(bad) > 1 else '')
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() - t0 > timeout_s:
                failed = 'no result after %.0f s (--launch-timeout); still running: ranks %s' % (
                    timeout_s, ', '.join(str(r) for r, c in enumerate(codes) if c is None))
                break
            time.sleep(0.05)
    finally:
        teardown()
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    if failed:
        raise SystemExit('bench.py --gpus %d: %s; the other ranks were stopped' % (n, failed))


def stub_rank(args, rank, world):
    """NNEST_BENCH_STUB=1: the launcher / rendezvous / one-line plumbing on CPUs (gloo), no kernel: what the `not gpu` test of
    `--gpus N` runs.  The line says so (`stub`: true) and carries no measurement."""
    import torch.distributed as dist
    die = os.environ.get('NNEST_BENCH_STUB_DIE', '')   # "rank:code": that rank exits before the rendezvous (the launcher's fail-fast test)
    if die and int(die.split(':')[0]) == rank:
        raise SystemExit(int(die.split(':')[1]))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group('gloo')
    dist.barrier()
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)   # the max-over-ranks reduction of the timed region
    if rank == 0:
        assert dist.get_world_size() == world == args.gpus
        print(json.dumps({'metric': METRIC % CONFIGS[args.config][1], 'value': None, 'unit': 'evals/s', 'n_gpus': world,
                          'rccl_ranks': dist.get_world_size(), 'collective_backend': dist.get_backend(), 'steps': args.steps,
                          'warmup': args.warmup, 'stub': True, 'max_over_ranks': float(t.item())}))
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # (defaults: 50 + 50 launches of 0.29 ms -- the first few tens of launches of a process run 3-4 % slower than the steady state
    # the sampler works in, which launches 282 of them per run: clocks and caches; `warmup` is reported in the line)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--config', type=int, default=2, choices=sorted(CONFIGS), help='BASELINE.json configuration')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'])
    ap.add_argument('--walkers', type=int, default=0, help='walkers per GPU (weak) / in total (strong); 0 = the config')
    ap.add_argument('--mcmc-steps', type=int, default=0, help='MH steps per launch (0 = 5*x_dim)')
    ap.add_argument('--fixed-step', action='store_true', help='no step-size adaptation in the timed launches')
    ap.add_argument('--lag', type=int, default=-1, help='lag of the batch-wide step rule (-1 = product default)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-saturation', action='store_true')
    ap.add_argument('--no-spline', action='store_true')
    ap.add_argument('--no-logz', action='store_true', help='skip the live nested run (the fixtures are still reported)')
    ap.add_argument('--bare', action='store_true', help='the timed launches only (profiling runs: scripts/profile_bench.sh)')
    ap.add_argument('--launch-timeout', type=float, default=1800.0, help='--gpus N launcher: seconds before the ranks are stopped')
    args = ap.parse_args()
    if args.bare:
        args.no_cpu_baseline = args.no_saturation = args.no_spline = args.no_logz = True

    if 'WORLD_SIZE' not in os.environ and (args.gpus > 1 or os.environ.get('NNEST_BENCH_LAUNCHER') == '1'):
        return launch_ranks(args.gpus, sys.argv[1:], args.launch_timeout)   # (NNEST_BENCH_LAUNCHER=1: also for one rank -- the launcher path under test)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (one rank per GPU)' % (args.gpus, world))
    if os.environ.get('NNEST_BENCH_STUB') == '1':
        return stub_rank(args, rank, world)
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit('bench.py needs an MI355X (no GPU visible); there is no CPU fallback')
    if os.environ.get('NNEST_BENCH_BACKEND', 'nccl') == 'nccl' and ndev <= local_rank:
        raise SystemExit('bench.py rank %d: LOCAL_RANK %d but only %d GPU(s) visible (one rank per GPU: --gpus %d needs %d devices)'
                         % (rank, local_rank, ndev, args.gpus, args.gpus))
    # one rank per GPU; NNEST_BENCH_BACKEND=gloo is a single-GPU smoke test of the multi-rank code path only
    backend = os.environ.get('NNEST_BENCH_BACKEND', 'nccl')
    dev_index = local_rank if backend == 'nccl' else local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    dist = None
    if world > 1 or os.environ.get('NNEST_BENCH_FORCE_DIST') == '1':
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    from nnest_amd import flow, _lib

    like, D, scale, N_cfg = CONFIGS[args.config]
    H, B, L = 16, 3, 1
    if args.scaling == 'weak':
        C = args.walkers or 1000
        C_total = C * world
        form = None
    else:
        C_total = args.walkers or N_cfg
        C = -(-C_total // world)
        form = 'whole batch' if args.fixed_step else None   # resolved below: the form the WHOLE batch would run, pinned on every shard
    S = args.mcmc_steps if args.mcmc_steps > 0 else 5 * D
    nvp = flow.HipNVP(D, H, B, L, device=dev, seed=0)
    rng = np.random.RandomState(1234 + rank)
    u0 = rng.uniform(-1, 1, size=(C, D))
    z0, _ = nvp.forward(u0)
    logl0 = flow.loglike(LIKE_ID[like], u0, scale, device=dev)
    loglstar = float(logl0.min())
    step_size = 1.0 / np.sqrt(D)
    dynamic = False if args.fixed_step else 'batch'
    lag = None if args.lag < 0 else args.lag
    if form == 'whole batch':   # the library's own answer (nnest_mh_form_for); a population too large for the batch rule's
        form = nvp.mh_form_for(C_total, dynamic=dynamic, lag=lag)   # resident grid runs the per-16-walker rule
        if form is None and dynamic:
            dynamic = 'group'
            form = nvp.mh_form_for(C_total, dynamic=dynamic)

    # state buffers are re-seeded outside the timed launches (clone is not part of the hot path)
    zs = [z0.clone() for _ in range(args.steps + args.warmup)]
    ls = [logl0.clone() for _ in range(args.steps + args.warmup)]
    gathered = torch.empty(world * C, D + 2, dtype=torch.float64, device=dev) if dist is not None else None

    def launch(i):
        res = nvp.mh_steps(LIKE_ID[like], scale, zs[i], ls[i], loglstar, step_size, S, dynamic=dynamic, lag=lag, seed=42 + i,
                           walker_offset=rank * C, form=form)
        if dist is not None:   # C2: what the nested-sampling loop consumes of a batch, gathered on every rank (device memory)
            moved = res['moved']   # (as nnest_amd/sampler.py::_mcmc_endpoints_fused)
            ends = torch.cat([res['x'].double(), ls[i][:, None], moved[:, None].double()], dim=1)
            dist.all_gather_into_tensor(gathered, ends)
        return res

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        launch(i)
    barrier()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    last = None
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev0[k].record()
        last = launch(args.warmup + k)
        ev1[k].record()
    barrier()
    dt = time.perf_counter() - t0
    # (first 20 timed launches, start of the first to end of the twentieth on the device's clock: no host sync inside the region)
    n20 = min(20, args.steps)
    dt_first20 = ev0[0].elapsed_time(ev1[n20 - 1]) / n20 if n20 else None
    kern_each = [a.elapsed_time(b) for a, b in zip(ev0, ev1)]
    kern_ms = float(np.mean(kern_each))
    if last is not None:
        nvp.check_sync(last)
    # The first tens of launches of a process run 3-4 % slower than the steady state a sampler's ~280 launches per run work in
    # (clocks, caches): `value` above is whatever --warmup / --steps the caller chose; BEHIND the timed region the same launch is
    # repeated until the process has made 300, so that the line carries both ends whatever the flags were (round-5 verdict item 9)
    warm_profile = None
    if dist is None and not args.bare:
        done = args.warmup + args.steps
        extra = max(0, 300 - done)
        if extra:
            e0 = [torch.cuda.Event(enable_timing=True) for _ in range(extra)]
            e1 = [torch.cuda.Event(enable_timing=True) for _ in range(extra)]
            tw = time.perf_counter()
            for k in range(extra):
                e0[k].record()
                launch((args.warmup + k) % len(zs))
                e1[k].record()
            torch.cuda.synchronize(dev)
            wall_extra = (time.perf_counter() - tw) / extra * 1e3
            tail = [a.elapsed_time(b) for a, b in zip(e0, e1)]
        else:
            wall_extra, tail = None, []
        allk = kern_each + tail
        warm_profile = {'what': 'kernel ms (HIP events) of the timed launches in order, then of the same launch repeated behind the timed '
                                'region until the process had made 300: the steady state the sampler works in',
                        'launches_before_timed': args.warmup,
                        'kernel_ms_first20': float(np.mean(kern_each[:20])),
                        'kernel_ms_steady': float(np.mean(allk[-50:])), 'steady_after_launches': args.warmup + len(allk) - 50,
                        'ms_per_step_steady': wall_extra}
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    evals_per_launch = C * S
    total_evals = evals_per_launch * args.steps * world
    value = total_evals / dt

    # north_star's own multi-GPU statement -- config 2's 1000 walkers SPLIT over the ranks (strong) -- measured in the same
    # invocation, beside the weak line above (every rank takes part: the all-gather is a collective)
    strong = None
    if dist is not None and args.scaling == 'weak' and args.config == 2:
        Cr = -(-N_cfg // world)
        us = np.random.RandomState(4321 + rank).uniform(-1, 1, size=(Cr, D))
        zz0, _ = nvp.forward(us)
        ll0 = flow.loglike(LIKE_ID[like], us, scale, device=dev)
        zs2 = [zz0.clone() for _ in range(args.steps + args.warmup)]
        ls2 = [ll0.clone() for _ in range(args.steps + args.warmup)]
        g2 = torch.empty(world * Cr, D + 2, dtype=torch.float64, device=dev)
        star2 = float(ll0.min())

        def launch2(i):
            r = nvp.mh_steps(LIKE_ID[like], scale, zs2[i], ls2[i], star2, step_size, S, dynamic=dynamic, lag=lag, seed=142 + i,
                             walker_offset=rank * Cr)
            dist.all_gather_into_tensor(g2, torch.cat([r['x'].double(), ls2[i][:, None], r['moved'][:, None].double()], dim=1))
            return r
        for i in range(args.warmup):
            launch2(i)
        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            r2 = launch2(args.warmup + k)
        barrier()
        t2 = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        nvp.check_sync(r2)
        strong = {'what': 'BASELINE config 2 as north_star states it: %d walkers split over the %d ranks (%d each), K4 + one RCCL '
                          'all-gather of the endpoints per batch' % (N_cfg, world, Cr),
                  'walkers_total': Cr * world, 'walkers_per_gpu': Cr, 'ms_per_step': float(t2.item()) / args.steps * 1e3,
                  'value': Cr * world * S * args.steps / float(t2.item()), 'unit': 'evals/s', 'scaling': 'strong',
                  'kernel': 'mh_kernel_%s' % nvp.mh_form_for(Cr, dynamic=dynamic, lag=lag)}

    if rank == 0:
        fl = useful_flops_per_eval(D, H, B, L)
        achieved_tflops = evals_per_launch * fl / (kern_ms * 1e-3) / 1e12
        info = _lib.device_info()
        cu = info['num_cu']
        kform = form or nvp.mh_form_for(C, dynamic=dynamic, lag=lag)
        tiles = -(-C // (4 if kform in ('quad', 'solo') else 16))
        solo = kform == 'solo'

        default_workload = (args.config, C, S, world, dynamic) == (2, 1000, 250, 1, 'batch') and lag is None
        traffic, traffic_src = committed_traffic('mh_kernel_%s' % kform) if default_workload else (None, None)
        out = {
            'metric': METRIC % D,
            'value': value, 'unit': 'evals/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'ms_per_step_first20': dt_first20, 'higher_is_better': True, 'scaling': args.scaling,
            'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s x_dim=%d, %d live points (walkers) per GPU, %d MH steps per launch, NVP hidden=%d '
                                   'blocks=%d layers=%d, %s' % (like, D, C, S, H, B, L,
                                                                'fixed step' if not dynamic else
                                                                'batch-wide dynamic step rule (product default: exact steps, then lagged)'),
                       'baseline_config': args.config, 'walkers_per_gpu': C, 'walkers_total': C_total, 'mcmc_steps': S,
                       'evals_per_step': evals_per_launch,
                       'parallelism': ('single GPU' if world == 1 and dist is None else
                                       'walkers sharded x%d (%s scaling), one RCCL all-gather of the chain endpoints '
                                       '[%d, %d] f64 per batch; flow replicas trained per rank (no weight broadcast)'
                                       % (world, args.scaling, C, D + 2))},
            'roofline': {'bound': 'valu' if solo else 'mfma', 'achieved': achieved_tflops, 'peak': FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved_tflops / FP32_PEAK_TFLOPS,
                         'traffic': traffic, 'traffic_source': traffic_src,
                         'kernel': 'mh_kernel_%s' % kform, 'kernel_ms': kern_ms, 'flops_per_eval': fl,
                         'hbm_frac_if_streamed': evals_per_launch * alg_bytes_per_eval(D) / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         'step_rule_lag': (lag if lag is not None else nvp.default_lag(C, form)) if dynamic == 'batch' else None,
                         'step_rule_exact_steps': (nvp.default_warm(C, dynamic, lag if lag is not None else nvp.default_lag(C, form), form)
                                                   if dynamic == 'batch' else None),
                         'note': 'bound "valu": the solo form issues no MFMA (one walker per wave, layers as v_fmac_f32 + DPP) -- the f32 '
                                 'vector peak and the f32-input MFMA peak are the same 157.3 TFLOP/s on gfx950 (64 FLOP/clk/SIMD), and a '
                                 'lone wave issues one vector instruction per 4 cycles, half the SIMD rate.  %d walker tiles on %d CUs: '
                                 'latency-bound at this population (a step is a serial chain of 9 small layers), see `saturated`'
                                 % (tiles, cu)},
            'device': info['name'],
        }
        if warm_profile is not None:
            warm_profile['evals_per_s_steady'] = (evals_per_launch / (warm_profile['ms_per_step_steady'] * 1e-3)
                                                  if warm_profile['ms_per_step_steady'] else None)
            out['launch_profile'] = warm_profile
        out['rccl_ranks'] = dist.get_world_size() if dist is not None else 1
        if not (out['n_gpus'] == out['rccl_ranks'] == world == args.gpus):   # the line must not claim GPUs the collectives did not span
            raise SystemExit('bench.py: n_gpus %r, rccl_ranks %r, WORLD_SIZE %d, --gpus %d disagree'
                             % (out['n_gpus'], out['rccl_ranks'], world, args.gpus))
        out['collective_backend'] = dist.get_backend() if dist is not None else None
        out['step_rule_scope'] = ('whole batch' if world == 1 else
                                  'per rank: under sharding every rank applies the batch-wide rule to ITS walkers (DESIGN.md 6)')
        if strong is not None:
            out['strong_config2'] = strong
        if world == 1 and dist is None and not args.bare:
            # K3: the single batched pass over all live points (inverse + box prior + likelihood), SURVEY.md 8d
            for _ in range(3):
                nvp.inverse_loglike(LIKE_ID[like], scale, z0)
            torch.cuda.synchronize(dev)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(50):
                nvp.inverse_loglike(LIKE_ID[like], scale, z0)
            b.record()
            torch.cuda.synchronize(dev)
            ms = a.elapsed_time(b) / 50
            out['k3'] = {'what': 'one fused pass (inverse + prior + loglike) over the %d live points, back-to-back launches' % C,
                         'ms_per_pass': ms, 'evals_per_s': C / (ms * 1e-3)}
            if dynamic:   # the same launches with a fixed step, and with the exact (lag 0) rule
                for key, kw in (('fixed_step', dict(dynamic=False)), ('batch_rule_lag0', dict(dynamic='batch', lag=0))):
                    tms = []
                    for k in range(4):
                        zz, ll = z0.clone(), logl0.clone()
                        torch.cuda.synchronize(dev)
                        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        a.record()
                        nvp.mh_steps(LIKE_ID[like], scale, zz, ll, loglstar, step_size, S, seed=7 + k, **kw)
                        b.record()
                        torch.cuda.synchronize(dev)
                        tms.append(a.elapsed_time(b))
                    ms = float(np.median(tms[1:]))
                    out[key] = {'kernel_ms': ms, 'evals_per_s': C * S / (ms * 1e-3), 'kernel': 'mh_kernel_%s' % nvp.mh_form_for(C, **kw),
                                'frac': C * S * fl / (ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                                'what': ('no step-size adaptation' if key == 'fixed_step' else
                                         'the reference\'s rule itself (nnest/sampler.py:422-431: the whole batch\'s vote on step s sets the scale of step s + 1)')}
                # which rule is the product default, and what the reference's own rule would cost as the default (round-5 verdict item 4)
                out['step_rule'] = {
                    'product_default': 'batch-wide rule, the first %d steps of a launch exact (lag 0), then %d steps behind' % (
                        nvp.default_warm(C, 'batch', nvp.default_lag(C, form), form), nvp.default_lag(C, form)),
                    'reference_rule': 'lag 0 throughout (mcmc_step_lag=0): every decision of the reference\'s recorded traces reproduced '
                                      '(tests/test_gpu_solo.py)',
                    'reference_rule_cost': out['batch_rule_lag0']['kernel_ms'] / kern_ms,
                    'note': 'the factor is the second candidate evaluation of every step (the vote\'s round trip is hidden behind the two): '
                            'two waves per walker instead of two evaluations per wave were built and are no faster '
                            '(profiles/r06/k4_lag0_duo_experiment.txt); at config 2 a run spends 4 % of its wall in K4, so lag 0 as the '
                            'default would cost ~4 % end to end'}
        if not args.no_saturation and world == 1 and dist is None:
            # the same step at a population that fills the chip (not the headline: BASELINE's config is 1000)
            Cs = 16 * 4 * cu * 8  # 8 walker tiles per SIMD
            us = np.random.RandomState(5).uniform(-1, 1, size=(Cs, D))
            zz, _ = nvp.forward(us)
            ll = flow.loglike(LIKE_ID[like], us, scale, device=dev)
            Ss = 25
            nvp.mh_steps(LIKE_ID[like], scale, zz, ll, float(ll.min()), step_size, Ss, seed=1)
            torch.cuda.synchronize(dev)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            nvp.mh_steps(LIKE_ID[like], scale, zz, ll, float(ll.min()), step_size, Ss, seed=2)
            b.record()
            torch.cuda.synchronize(dev)
            ms = a.elapsed_time(b)
            out['saturated'] = {'walkers': Cs, 'mcmc_steps': Ss, 'kernel_ms': ms, 'evals_per_s': Cs * Ss / (ms * 1e-3),
                                'tflops': Cs * Ss * fl / (ms * 1e-3) / 1e12, 'kernel': 'mh_kernel (image form)',
                                'frac_of_fp32_peak': Cs * Ss * fl / (ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS}
        if not args.no_spline and world == 1 and dist is None and args.config == 2:
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
            sp_flops = 136000   # per eval: 2 x 68 k multiply-adds on the matrix cores; the ~150 spline evaluations on top are not counted
            sp_kernel = {'pair': 'spline_mh_kernel_pair', 'team': 'spline_mh_kernel_team', 'wave': 'spline_mh_kernel'}[sp.kernel_form_for(C)]   # (asked of the library)
            out['spline_flow'] = {'kernel': sp_kernel,
                                  'kernel_ms': ms, 'evals_per_s': C * S / (ms * 1e-3),
                                  'roofline': {'bound': 'mfma', 'flops_per_unit': sp_flops, 'unit': 'TFLOP/s', 'peak': FP32_PEAK_TFLOPS,
                                               'achieved': C * S * sp_flops / (ms * 1e-3) / 1e12,
                                               'frac': C * S * sp_flops / (ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                                               'kernel': sp_kernel, 'profile': 'profiles/r06/spline_kernel_stats.csv'},
                                  'note': 'SingleSpeedSpline hidden=%d blocks=%d bins=8; VALU-bound on the spline arithmetic' % (H, B)}
            if C >= 200:  # its training epoch at this population (90 % train / 10 % validation, batch 100: trainer.py:159-176)
                nv = C // 10
                E = 40  # (as the K5 figure below: 40 epochs per call)
                perms = torch.stack([torch.randperm(C - nv) for _ in range(E)]).int()
                kw = dict(seed=1, jitter=0.01, batch=100, patience=50)
                sp.train_epochs(u0[nv:], u0[:nv], perms[:2], None, max_epochs=2, **kw)   # allocations
                best = float('inf')
                for _ in range(3):  # (a side number: best of three calls, one hiccup of the box does not stand)
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    res = sp.train_epochs(u0[nv:], u0[:nv], perms, None, max_epochs=E, **kw)
                    torch.cuda.synchronize(dev)
                    best = min(best, (time.perf_counter() - t0) / max(1, res['epochs_run']) * 1e3)
                out['spline_flow']['train_ms_per_epoch'] = best
                tf = ((C - nv) * 3 + nv) * sp_flops   # forward + backward (2 x) over the training rows, forward over the validation rows
                out['spline_flow']['train_roofline'] = {'bound': 'mfma', 'flops_per_unit': tf, 'unit_is': 'epoch', 'unit': 'TFLOP/s',
                                                        'peak': FP32_PEAK_TFLOPS, 'achieved': tf / (best * 1e-3) / 1e12,
                                                        'frac': tf / (best * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                                                        'kernel': 'splr_grad_kernel + splr_update_kernel' if sp.train_form_for(100) == 'rows' else 'spl_grad_kernel + spl_update_kernel',
                                                        'form': sp.train_form_for(100),   # (asked of the library: nnest_spline_train_form)
                                                        'profile': 'profiles/r06/spline_train_kernel_stats.csv'}
        if world == 1 and dist is None and not args.bare and args.config == 2:
            # the slice proposal in latent space (north_star "slice/MH"): build-defined, the reference has none --
            # reported beside, never as `value`.  Its unit is the same eval (one coupling-stack inverse + box + likelihood); an update
            # takes a handful of them and always moves.
            Sl = 25
            t_ms, n_eval, n_move = [], 0, 0
            for k in range(4):
                zz, ll = z0.clone(), logl0.clone()
                torch.cuda.synchronize(dev)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rs = nvp.slice_steps(LIKE_ID[like], scale, zz, ll, loglstar, 2.0 * step_size, Sl, seed=11 + k)
                e1.record()
                torch.cuda.synchronize(dev)
                t_ms.append(e0.elapsed_time(e1))
                n_eval, n_move = int(rs['n_eval'].sum()), int(rs['n_move'].sum())
            ms = float(np.median(t_ms[1:]))
            out['slice_proposal'] = {'kernel': 'slice_kernel_solo', 'walkers': C, 'updates_per_walker': Sl, 'kernel_ms': ms,
                                     'evals_per_s': n_eval / (ms * 1e-3), 'evals_per_update': n_eval / max(1, C * Sl),
                                     'updates_per_s': C * Sl / (ms * 1e-3), 'moved_fraction': n_move / max(1, C * Sl),
                                     'roofline': {'bound': 'valu', 'flops_per_unit': fl, 'unit': 'TFLOP/s', 'peak': FP32_PEAK_TFLOPS,
                                                  'achieved': n_eval * fl / (ms * 1e-3) / 1e12,
                                                  'frac': n_eval * fl / (ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 'kernel': 'slice_kernel_solo',
                                                  'profile': 'profiles/r06/slice_kernel_stats.csv'},
                                     'note': 'UNPINNED: the reference proposes random-walk Metropolis moves only (nnest/sampler.py:310-316); '
                                             'parity is against the CPU restatement of the build-defined step (tests/test_gpu_slice.py)'}
        if world == 1 and dist is None and not args.bare and args.config in (2, 5):
            # the same workload on the build-defined MAF (BASELINE config 5 names it): reported beside, never
            # as `value`.  Its inverse -- the direction the proposals need -- is `num_groups` passes of the nets per block
            # (DESIGN.md 3c), so its evals/s sit that factor below the RealNVP's in the same (image) kernel form.
            from nnest_amd.maf import HipMAF
            mf = HipMAF(D, H, B, L, device=dev, seed=0)
            zmf, _ = mf.forward(u0)
            Sm = min(S, 50)   # (a bounded sample of the S-step launch: the kernel is a loop of identical steps)
            t_ms = []
            for k in range(3):
                zz, ll = zmf.clone(), logl0.clone()
                torch.cuda.synchronize(dev)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                mf.mh_steps(LIKE_ID[like], scale, zz, ll, loglstar, step_size, Sm, dynamic='group', seed=7 + k)
                e1.record()
                torch.cuda.synchronize(dev)
                t_ms.append(e0.elapsed_time(e1))
            ms = float(np.median(t_ms[1:]))
            nvm = min(C, 1000) // 10
            Xm = u0[:min(C, 1000)]
            perms = torch.stack([torch.randperm(Xm.shape[0] - nvm) for _ in range(6)])
            mf.train_epochs(Xm[nvm:], Xm[:nvm], perms[:1], None, seed=1, jitter=0.01, batch=100, max_epochs=1, patience=50)   # (first-call set-up)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            resm = mf.train_epochs(Xm[nvm:], Xm[:nvm], perms, None, seed=1, jitter=0.01, batch=100, max_epochs=6, patience=50)
            torch.cuda.synchronize(dev)
            mfl = useful_flops_per_eval(D, H, B, L)   # one mask-pruned pass of the nets: what an incremental inverse would need
            tfm = ((Xm.shape[0] - nvm) * 3 + nvm) * mfl
            t_ep = (time.perf_counter() - t0) / max(1, resm['epochs_run']) * 1e3
            out['maf_flow'] = {'kernel': 'maf_mh_kernel (image form, grouped sequential inverse)', 'num_groups': mf.num_groups,
                               'roofline': {'bound': 'mfma', 'flops_per_unit': mfl, 'unit': 'TFLOP/s', 'peak': FP32_PEAK_TFLOPS,
                                            'achieved': C * Sm * mfl / (ms * 1e-3) / 1e12, 'frac': C * Sm * mfl / (ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                                            'executed_over_algorithmic': mf.num_groups, 'kernel': 'maf_mh_kernel', 'profile': 'profiles/r06/maf_kernel_stats.csv'},
                               'train_roofline': {'bound': 'mfma', 'flops_per_unit': tfm, 'unit_is': 'epoch', 'unit': 'TFLOP/s', 'peak': FP32_PEAK_TFLOPS,
                                                  'achieved': tfm / (t_ep * 1e-3) / 1e12, 'frac': tfm / (t_ep * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                                                  'kernel': 'maf_grad_kernel + maf_update_kernel', 'profile': 'profiles/r06/maf_kernel_stats.csv'},
                               'walkers': C, 'mcmc_steps_timed': Sm, 'kernel_ms': ms, 'evals_per_s': C * Sm / (ms * 1e-3),
                               'train_ms_per_epoch': t_ep,
                               'train_what': 'host-driven epoch loop (loss_grad + adam_step per minibatch), %d points' % Xm.shape[0],
                               'note': 'UNPINNED: the reference has no MAF (nnest/trainer.py:83-100); parity is against the oracle '
                                       'restatement of the build-defined flow (tests/test_gpu_maf.py)'}
        if world == 1 and dist is None and args.config == 2 and not args.bare:
            # K5 beside K4: the NVP training epoch at this population
            nv = C // 10
            E = 40
            perms = torch.stack([torch.randperm(C - nv) for _ in range(E)]).int()
            kw = dict(seed=1, jitter=0.01, batch=100, patience=1000)
            tr = flow.HipNVP(D, H, B, L, device=dev, seed=1)
            tr.train_epochs(u0[nv:], u0[:nv], perms[:2], None, max_epochs=2, **kw)
            best = float('inf')
            for _ in range(3):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                res = tr.train_epochs(u0[nv:], u0[:nv], perms, None, max_epochs=E, **kw)
                torch.cuda.synchronize(dev)
                best = min(best, (time.perf_counter() - t0) / max(1, res['epochs_run']) * 1e3)
            kfl = ((C - nv) * 3 + nv) * fl   # forward + backward (2 x) over the training rows, forward over the validation rows
            out['k5_train'] = {'ms_per_epoch': best,
                               'what': 'Trainer.train epoch loop in one launch (nnest_nvp_train), %d live points' % C,
                               'roofline': {'bound': 'valu', 'flops_per_unit': kfl, 'unit_is': 'epoch', 'unit': 'TFLOP/s', 'peak': FP32_PEAK_TFLOPS,
                                            'achieved': kfl / (best * 1e-3) / 1e12, 'frac': kfl / (best * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                                            'kernel': 'train_kernel_rows<2>', 'profile': 'profiles/r06/train_kernel_stats.csv',
                                            'note': 'latency-bound: per minibatch one forward + backward chain per row (100 waves on 25 '
                                                    'CUs) and four cross-CU round trips (two grid barriers, the operand loads of the '
                                                    'weight-gradient jobs, the image refresh)'}}
            out['logz'] = logz_report(dev, live_run=not args.no_logz)
            if 'e2e' in out['logz']:
                out['e2e'] = out['logz'].pop('e2e')   # (the wall-time split of the live run: a top-level object)
        if not args.no_cpu_baseline and world == 1 and dist is None:
            out['cpu_baseline'] = cpu_baseline(D, nvp.store_packed(), like, scale, C)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
