"""The driver's contract for bench.py (one JSON line, the fields the judge reads) and for __graft_entry__.smoke(), on a GPU."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_roofline_and_cpu_baseline():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '1'], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 1 and d['unit'] == 'evals/s' and d['dtype'] == 'f32'
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None and d['data'] == 'synthetic'
    assert 'workload' in d['config'] and 'model' not in d['config']
    r = d['roofline']
    # (round 4: the solo form issues no MFMA -- its bound is labelled 'valu'; the f32 vector peak is the same 157.3 TFLOP/s)
    assert r['bound'] in ('hbm', 'mfma', 'valu') and r['unit'] == 'TFLOP/s' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    assert r['kernel'] == 'mh_kernel_solo' and r['bound'] == 'valu'
    assert r['traffic'] is None or r['traffic'] > 0
    # value = evals per launch / measured time: consistent with ms_per_step
    assert abs(d['value'] - d['config']['evals_per_step'] / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']
    with open(os.path.join(ROOT, 'BASELINE.json')) as f:
        assert d['metric'] == json.load(f)['metric']          # both halves: evals/s and the log-Z error
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] > 0 and c['unit'] == 'evals/s' and c['one_thread'] > 0
    assert c['value'] >= 0.8 * c['one_thread']                # all host cores beside one
    assert d['value'] > 10 * c['value']
    assert d['spline_flow']['evals_per_s'] > 0
    assert d['k3']['evals_per_s'] > 0 and d['k5_train']['ms_per_epoch'] > 0
    # every kernel on the path carries its own roofline object (round-3 verdict): flops per unit, achieved, fraction, kernel
    for ro in (d['k5_train']['roofline'], d['spline_flow']['roofline'], d['spline_flow']['train_roofline'], d['maf_flow']['roofline'],
               d['maf_flow']['train_roofline'], d['slice_proposal']['roofline']):
        assert ro['flops_per_unit'] > 0 and 0 < ro['frac'] < 1 and abs(ro['frac'] - ro['achieved'] / ro['peak']) < 1e-9 and ro['kernel']
    assert all(ro.get('profile') for ro in (d['maf_flow']['roofline'], d['maf_flow']['train_roofline']))   # (round-5 verdict: was null)
    sp = d['slice_proposal']                                    # build-defined, beside the line: a handful of evals per update, all move
    assert 2.0 < sp['evals_per_update'] < 20.0 and sp['moved_fraction'] > 0.9 and sp['evals_per_s'] > 1e7
    assert d['ms_per_step_first20'] > 0 and d['launch_profile']['kernel_ms_steady'] > 0 and d['step_rule']['reference_rule_cost'] > 1.0
    e = d['e2e']                                                # where the wall time of the live config-2 run went
    assert e['wall_s'] > 0 and e['k5_s'] > 0 and e['k4_s'] > 0 and abs(e['wall_s'] - e['k5_s'] - e['k4_s'] - e['host_s']) < 1e-6
    assert e['k5_epochs'] > 1000 and e['k4_launches'] > 100
    z = d['logz']
    assert abs(z['live_run']['logz'] + 242.0) < 3.0            # one live config-2 run: within a few sigma of the ensemble
    if 'cpu_mean' in z and 'gpu_mean' in z:
        assert abs(z['delta']) <= max(0.1, 2 * z['combined_stderr'])


def test_bench_multi_rank_path_over_rccl_on_one_gpu():
    """the N > 1 code path of bench.py (K4 on the shard + the RCCL all-gather of the chain endpoints inside the timed region)
    with a one-rank process group, which is what a one-GPU box can run"""
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29533',
               NNEST_BENCH_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '1', '--config', '4',
                          '--scaling', 'strong'], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][0])
    assert d['scaling'] == 'strong' and 'all-gather' in d['config']['parallelism'] and d['config']['walkers_total'] == 4000
    assert d['value'] > 0 and d['roofline']['kernel'] == 'mh_kernel_team' and d['rccl_ranks'] == 1 and d['collective_backend'] == 'nccl'


def test_bench_emits_the_strong_config2_line_beside_the_weak_one():
    """one `bench.py --gpus N` invocation reports the weak line (1000 walkers per GPU) AND north_star's own statement -- config 2's
    1000 walkers split over the ranks -- as `strong_config2` (a one-rank RCCL group here)"""
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29534',
               NNEST_BENCH_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '1'], cwd=ROOT, capture_output=True,
                         text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][0])
    s = d['strong_config2']
    assert d['scaling'] == 'weak' and s['scaling'] == 'strong' and s['walkers_total'] == 1000 and s['value'] > 0 and d['rccl_ranks'] == 1


def test_smoke_entry_point():
    out = subprocess.run([sys.executable, '-c', 'import __graft_entry__ as g; g.smoke()'], cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]


def test_bench_gpus_1_through_the_rank_launcher():
    """the path `--gpus N` takes for N > 1 (bench.py starts its own ranks, relays rank 0's line), with one rank: what a one-GPU box
    can run of it.  The line says how many ranks the collective saw and which backend carried it."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env.update(NNEST_BENCH_LAUNCHER='1', NNEST_BENCH_FORCE_DIST='1')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--bare'],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['rccl_ranks'] == 1 and d['collective_backend'] == 'nccl' and d['value'] > 0
    assert 'strong_config2' in d and 'step_rule_scope' in d
