"""world_size-2 tests of the multi-process path on CPU (gloo): replicated evidence state, MCMC batch sharded
over ranks, endpoints all-gathered (C2), rank-0 retrain + weight broadcast (C3).  The flow arithmetic comes
from the TEST-ONLY oracle-backed trainer; what is under test is nnest_amd.sampler / nnest_amd.nested."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp, out):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from nnest_amd.nested import NestedSampler
        from nnest_amd.likelihoods import Rosenbrock
        from tests.oracle_trainer import OracleTrainer
        np.random.seed(100 + rank)      # deliberately different per rank: rank 0's draws must win
        torch.manual_seed(100 + rank)
        tr = OracleTrainer(2, seed=7)   # same initial weights on every rank
        rosen = Rosenbrock(2)

        def like(x):   # with one derived column: it has to travel with its point through both collectives
            return rosen(x), (x[:, :1] + x[:, 1:2])

        s = NestedSampler(2, like, transform=lambda x: 5 * x, log_dir=tmp, num_live_points=200, trainer=tr,
                          log_level=40, num_derived=1)
        assert s.use_mpi and s.mpi_size == world and s.mpi_rank == rank
        # collective helpers
        g = s._all_gather_rows(np.full((3, 2), float(rank)))
        assert g.shape == (3 * world, 2) and np.array_equal(g[:3], np.zeros((3, 2))) and np.all(g[3:6] == 1)
        b = s._broadcast(np.arange(5) + 10.0 * rank)
        assert np.array_equal(b, np.arange(5))
        assert s._all_sum(rank + 1) == sum(range(1, world + 1))
        s.run(train_iters=100, mcmc_num_chains=9, mcmc_dynamic_step_size=False)  # 9 chains over 2 ranks: padded shard
        assert np.allclose(s.samples[:, 2], s.samples[:, 0] + s.samples[:, 1], atol=1e-5)
        w = tr.netG.store_packed()
        out.put((rank, float(s.logz), int(s.niter), int(s.ncall), float(np.sum(s.samples)), float(np.sum(w)),
                 tr.num_trains, s.logs is not None))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_nested_run_is_replicated(tmp_path):
    world = 2
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=500) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, logz0, nit0, ncall0, ssum0, wsum0, nt0, has_logs0), (r1, logz1, nit1, ncall1, ssum1, wsum1, nt1, has_logs1) = res
    # replicated state: every rank ends with the identical evidence, iteration count, samples and flow weights
    assert logz0 == logz1 and nit0 == nit1 and ncall0 == ncall1 and ssum0 == ssum1 and wsum0 == wsum1
    assert abs(logz0 + 5.80) <= 0.45   # 200 live points: sqrt(h/N) ~ 0.16
    assert nt0 >= 1 and nt1 == 0       # only rank 0 trains; the weights reach rank 1 by broadcast
    assert has_logs0 and not has_logs1  # only the primary process writes the run directory
