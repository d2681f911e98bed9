"""K5 (NVP training) time per epoch at a BASELINE population, multi-CU kernel and single-workgroup kernel (developer diagnostic).
  python tools/time_train.py [x_dim] [n_live]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnest_amd import flow  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 50
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rng = np.random.RandomState(0)
live = rng.uniform(-1, 1, size=(N, D))
nv = N // 10
E = 40
perms = torch.stack([torch.randperm(N - nv) for _ in range(E)]).int()
FORM = 'train_kernel_pipe' if os.environ.get('NNEST_TRAIN_FORM', '') == 'pipe' else 'train_kernel_rows'
for name, one_cu in (('multi-CU (%s)' % FORM, False), ('one CU   (train_kernel)', True)):
    nvp = flow.HipNVP(D, 16, 3, 1, seed=1)
    kw = dict(seed=1, jitter=0.01, batch=100, patience=1000, one_cu=one_cu)
    nvp.train_epochs(live[nv:], live[:nv], perms[:2], None, max_epochs=2, **kw)
    torch.cuda.synchronize()
    ts = []
    for k in range(3):
        t0 = time.perf_counter()
        res = nvp.train_epochs(live[nv:], live[:nv], perms, None, max_epochs=E, **kw)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / res['epochs_run'] * 1e3)
    print('x_dim %d, %d live points, %-32s %.3f ms per epoch (%d minibatches: %.1f us each)' % (
        D, N, name, min(ts), (N - nv + 99) // 100, min(ts) * 1e3 / ((N - nv + 99) // 100)))

if os.environ.get('NNEST_HIP_LIB', '').endswith('STAMP.so'):   # NNEST_STAMP build: cycles per phase of a minibatch (workgroup 0)
    nvp = flow.HipNVP(D, 16, 3, 1, seed=1)
    res = nvp.train_epochs(live[nv:], live[:nv], perms, None, max_epochs=E, seed=1, jitter=0.01, batch=100, patience=1000)
    ph = res['losses'].cpu().numpy().ravel()[:8] / (E * ((N - nv + 99) // 100))
    if os.environ.get('NNEST_TRAIN_FORM', '') != 'pipe':   # (the default form: train_kernel_rows)
        print('rows kernel, cycles per minibatch: forward %d  backward+staging %d  first grid barrier %d  weight-gradient jobs + Adam + publish %d  '
              'workgroup barrier %d  loss + image refresh (tag polls) %d  [first barrier: drain + workgroup barrier %d cycles, %.2f missed polls]' % tuple(ph[:8]))
    else:
        tl = res['losses'].cpu().numpy().ravel()[16:16 + 24]
        t0 = tl[0]
        rel = [int((x - t0) % (1 << 24)) if x else -1 for x in tl]
        print('pipe kernel, minibatch 20, cycles since the row wave entered block 0 forward: row wave 0: forward done %d, arrive(2) %d, arrive(1) %d, arrive(0) %d | '
              'owner of a block-0 job: arrivals seen %d, contracted %d, published %d | a re-laying wave: hints seen (block 2, 1, 0) %d %d %d, blocks in the images %d %d %d'
              % (rel[1], rel[2], rel[3], rel[4], rel[8], rel[9], rel[10], rel[12], rel[13], rel[14], rel[15], rel[16], rel[17]))
        print('   owner of a block-2 job (service wave 1): top of its loop %d, arrivals seen %d, contracted %d, published %d; re-laying wave: top of its loop %d; row wave 0 done with its share of the refresh %d' % (rel[18], rel[5], rel[6], rel[7], rel[19], rel[20]))
        # train_kernel_pipe (opt-in: NNEST_TRAIN_FORM=pipe; the default form is train_kernel_rows): row wave 0 and service wave 0 of workgroup 0 (the owner of a block-0 job)
        print('pipe kernel, cycles per minibatch, row wave: wait for rows + block 0 images %d  forward (+ staging, waits for blocks 1, 2) %d  '
              'backward to the last store %d  drain in front of the last arrival %d  | sum %d' % (tuple(ph[:4]) + (ph[:4].sum(),)))
        print('pipe kernel, cycles per minibatch, service wave 0: row preparation %d  wait for its block\'s arrivals %d  job + Adam + publish %d  '
              'refresh %d  | sum %d' % (tuple(ph[4:8]) + (ph[4:8].sum(),)))
