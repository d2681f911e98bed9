"""ORACLE-SIDE GENERATOR -- TEST INFRASTRUCTURE ONLY.  Produces the CPU-path evidence fixtures
tests/golden/logz_cpu_cfg{1,2,3}.json: the host driver (nnest_amd.nested.NestedSampler, host protocol, fused=False)
run end to end on the oracle-backed trainer (tests/oracle_trainer.py: the C restatement of the reference's flow,
training and nothing else), i.e. the reference's algorithm on the CPU for a BASELINE configuration and a list of seeds.

The reference as shipped cannot produce these numbers itself in a usable time (~59 likelihood calls/s
end to end -> ~10 days for config 2), which is why the CPU path is its restatement.  The proposal loop is the
reference's protocol loop (sampler.py:264-463) with its batch-wide dynamic step rule (sampler.py:422-431).

  python oracle/run_logz_cpu.py one <cfg> <seed> <out.json>          one run (20-40 CPU-minutes for config 2)
  python oracle/run_logz_cpu.py all <cfg> <seed,seed,...> [jobs]     runs in parallel, writes tests/golden/logz_cpu_cfg<cfg>.json
  python oracle/run_logz_cpu.py pool <cfg> <lo> <hi> <jobs> <dir>    seeds lo..hi-1, one <dir>/cfg<cfg>_seed<k>.json each (kept; done seeds skipped)
  python oracle/run_logz_cpu.py merge <cfg> <dir>                    fold every <dir>/cfg<cfg>_seed*.json into the fixture
"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (likelihood, x_dim, prior scale, live points[, mcmc_num_chains: the live-point count when absent])
# 11 = config 1 at the reference's DEFAULT mcmc_num_chains = 10 (nested.py:185) instead of one chain per live point
CONFIGS = {1: ('Rosenbrock', 2, 5.0, 100), 2: ('Rosenbrock', 50, 5.0, 1000), 3: ('GaussianMix', 20, 10.0, 2000),
           11: ('Rosenbrock', 2, 5.0, 100, 10)}


def config(cfg):
    c = CONFIGS[cfg]
    return c[0], c[1], c[2], c[3], (c[4] if len(c) > 4 else c[3])


def one(cfg, seed, out):
    os.environ.setdefault('OMP_NUM_THREADS', '1')
    import numpy as np
    import torch
    torch.set_num_threads(1)
    # Adam with weight decay drives the weights the parity mask never reaches to denormal values within ~90 epochs (update
    # ~ 0.1 w per step once |w| << 1e-2); x86 then runs every product with them ~100x slower (measured: 39 ms instead of 7.8 ms
    # per 1000-row inverse).  Those products are multiplied by exact zeros of the mask, so flushing denormals (MXCSR FTZ/DAZ,
    # this thread: the C oracle runs on it) changes no value that reaches the flow's output.
    torch.set_flush_denormal(True)
    from nnest_amd import likelihoods
    from nnest_amd.nested import NestedSampler
    from tests.oracle_trainer import OracleTrainer
    name, D, scale, N, chains = config(cfg)
    like = getattr(likelihoods, name)(D)
    np.random.seed(seed)
    torch.manual_seed(seed)
    tr = OracleTrainer(D, seed=seed)
    s = NestedSampler(D, like, transform=lambda x: scale * x, log_dir=tempfile.mkdtemp(dir='/tmp'), num_live_points=N,
                      trainer=tr, log_level=30, fused=False, flow='nvp')
    t0 = time.time()
    s.run(mcmc_num_chains=chains)     # the same call as the GPU runs it is compared with (tools/run_config.py)
    res = dict(config=cfg, seed=seed, likelihood=name, x_dim=D, num_live_points=N, mcmc_num_chains=chains, logz=float(s.logz),
               logzerr=float(s.logzerr), h=float(s.h), niter=int(s.niter), ncall=int(s.ncall), retrains=int(s.num_retrains),
               batches=int(s.num_batches), train_epochs_total=int(tr.total_iters), wall_s=time.time() - t0)
    with open(out, 'w') as f:
        json.dump(res, f)
    print(json.dumps(res))


def pool(cfg, seeds, jobs, outdir):
    """run the seeds `jobs` at a time; every finished seed leaves its own file, so an interrupted pool loses nothing"""
    os.makedirs(outdir, exist_ok=True)
    out = lambda sd: os.path.join(outdir, 'cfg%d_seed%d.json' % (cfg, sd))
    pend = [sd for sd in seeds if not os.path.exists(out(sd))]
    running = []
    while pend or running:
        while pend and len(running) < jobs:
            sd = pend.pop(0)
            running.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), 'one', str(cfg), str(sd), out(sd)],
                                            stdout=subprocess.DEVNULL))
        time.sleep(5)
        running = [p for p in running if p.poll() is None]
    return [out(sd) for sd in seeds]


def all_(cfg, seeds, jobs):
    outs = pool(cfg, seeds, jobs, tempfile.mkdtemp(dir='/tmp'))
    merge(cfg, [json.load(open(o)) for o in outs if os.path.exists(o)])


def merge(cfg, runs):
    import numpy as np
    path = os.path.join(ROOT, 'tests', 'golden', 'logz_cpu_cfg%d.json' % cfg)
    if os.path.exists(path):   # add to the seeds already there
        with open(path) as f:
            old = json.load(f)['runs']
        runs = [r for r in old if r['seed'] not in [q['seed'] for q in runs]] + runs
        runs.sort(key=lambda r: r['seed'])
    z = np.array([r['logz'] for r in runs])
    name, D, scale, N, chains = config(cfg)
    doc = dict(what='CPU-path log Z: host driver + oracle-backed trainer (oracle/run_logz_cpu.py)', config=cfg,
               likelihood=name, x_dim=D, num_live_points=N, mcmc_num_chains=chains, flow='nvp h16 b3 l1', train_iters=500,
               seeds=[r['seed'] for r in runs], logz=[r['logz'] for r in runs], mean=float(z.mean()),
               std=float(z.std(ddof=1)) if len(z) > 1 else None,
               stderr=float(z.std(ddof=1) / np.sqrt(len(z))) if len(z) > 1 else None, runs=runs)
    with open(path, 'w') as f:
        json.dump(doc, f, indent=1)
    print(path, doc['mean'], doc['std'])


if __name__ == '__main__':
    if sys.argv[1] == 'one':
        one(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
    elif sys.argv[1] == 'pool':
        pool(int(sys.argv[2]), range(int(sys.argv[3]), int(sys.argv[4])), int(sys.argv[5]), sys.argv[6])
    elif sys.argv[1] == 'merge':
        import glob
        merge(int(sys.argv[2]), [json.load(open(f)) for f in sorted(glob.glob(os.path.join(sys.argv[3], 'cfg%d_seed*.json' % int(sys.argv[2]))))])
    else:
        all_(int(sys.argv[2]), [int(v) for v in sys.argv[3].split(',')], int(sys.argv[4]) if len(sys.argv) > 4 else 4)
