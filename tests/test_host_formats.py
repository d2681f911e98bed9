"""On-disk products against what the REFERENCE wrote (tests/golden/formats/, captured from its seeded config-1 run by
oracle/gen_golden.py formats): text formats byte for byte, file and key names, and a run of this build resumed from a
reference-written checkpoint set (nnest/sampler.py:494-511; nnest/nested.py:92-95, :172-196, :473-485, :503-506)."""
import csv
import io
import json
import os

import numpy as np
import torch

from nnest_amd.likelihoods import Rosenbrock
from nnest_amd.nested import NestedSampler
from tests.oracle_trainer import OracleTrainer

F = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'formats')


def _sampler(tmp, **kw):
    return NestedSampler(2, Rosenbrock(2), transform=lambda x: 5 * x, log_dir=str(tmp), num_live_points=100, log_level=40,
                         trainer=OracleTrainer(2, seed=1), **kw)


def test_chain_txt_is_byte_identical_in_format(tmp_path):
    """chain.txt: 'weight minusloglike params...' rows, %.5E, weights floored at 1e-30, no header without param_names"""
    with open(os.path.join(F, 'chain_head_tail.txt')) as f:
        ref_lines = [ln for ln in f.read().split('\n') if ln and ln != '...']
    rows = np.array([[float(v) for v in ln.split()] for ln in ref_lines])
    s = _sampler(tmp_path)
    # weights below the floor come back as the floor: feed 0 for those to exercise the clamp
    weights = np.where(rows[:, 0] <= 1e-30, 0.0, rows[:, 0])
    s._save_samples(rows[:, 2:], -rows[:, 1], weights=weights)
    with open(os.path.join(s.logs['chains'], 'chain.txt')) as f:
        ours = f.read().split('\n')
    assert ours[:len(ref_lines)] == ref_lines and ours[len(ref_lines):] == ['']


def test_csv_headers_rows_and_file_names(tmp_path):
    with open(os.path.join(F, 'meta.json')) as f:
        meta = json.load(f)
    np.random.seed(3)
    torch.manual_seed(3)
    s = _sampler(tmp_path, checkpoint_min_seconds=0.0)
    s.run(train_iters=50, mcmc_num_chains=10)
    # run directory layout and checkpoint file names
    # (data/ and models/ are made by the Trainer, trainer.py:108-117; the injected test trainer has no directory)
    ours_layout = set(os.listdir(s.logs['run_dir']))
    assert {'chains', 'checkpoint', 'info', 'plots', 'results'} <= ours_layout <= set(meta['run_dir_layout'])
    names = set(os.listdir(s.logs['checkpoint']))
    for stem in ('active_u_%d.npy', 'active_v_%d.npy', 'active_logl_%d.npy', 'active_derived_%d.npy', 'checkpoint_%d.txt'):
        assert stem % 0 in names and stem % 20 in names          # log_interval = 0.2 * 100, as in the reference's set
    assert {'saved_v.npy', 'saved_logl.npy', 'saved_logwt.npy'} <= names
    assert [int(n.split('_')[1].split('.')[0]) for n in names if n.startswith('checkpoint_')].count(40) == 1
    # checkpoint JSON: the same keys, JSON-serialisable values of the same types
    last = max(int(n.split('_')[1].split('.')[0]) for n in names if n.startswith('checkpoint_'))
    with open(os.path.join(s.logs['checkpoint'], 'checkpoint_%d.txt' % last)) as f:
        ours = json.load(f)
    ref = meta['checkpoint_state']
    assert list(ours.keys()) == list(ref.keys())
    assert all(type(ours[k]) is type(ref[k]) for k in ref)
    # results.csv: header byte for byte; rows have the reference's ten columns and parse as its rows do
    with open(os.path.join(F, 'results_head.csv')) as f:
        ref_res = [ln for ln in f.read().split('\n') if ln and ln != '...']
    with open(os.path.join(s.logs['results'], 'results.csv')) as f:
        our_res = f.read().split('\n')
    assert our_res[0] == ref_res[0]
    ref_row = next(csv.reader(io.StringIO(ref_res[1])))
    our_row = next(csv.reader(io.StringIO(our_res[1])))
    assert len(our_row) == len(ref_row) == 10 and int(our_row[0]) % 20 == 0 and float(our_row[-1]) == int(our_row[-1])
    # final.csv: header byte for byte, one row of five numbers; niter and ncall written as integers like the reference's
    with open(os.path.join(F, 'final.csv')) as f:
        ref_fin = f.read().split('\n')
    with open(os.path.join(s.logs['results'], 'final.csv')) as f:
        our_fin = f.read().split('\n')
    assert our_fin[0] == ref_fin[0] and len(our_fin[1].split(',')) == 5
    assert our_fin[1].split(',')[0].isdigit() and our_fin[1].split(',')[1].isdigit() and ref_fin[1].split(',')[0].isdigit()
    # info/params.txt: a JSON object of strings; every key this build writes is one the reference writes
    with open(os.path.join(s.logs['info'], 'params.txt')) as f:
        params = json.load(f)
    assert set(params) <= set(meta['params_keys']) and all(isinstance(v, str) for v in params.values())


def test_resume_from_a_reference_written_checkpoint(tmp_path):
    """the reference's checkpoint set (its last one, iteration 680 of 691) laid out as it writes it; this build resumes from it
    and finishes the run: the evidence accumulated by the reference is carried over, the result lands on the reference's own."""
    with open(os.path.join(F, 'meta.json')) as f:
        meta = json.load(f)
    cps = np.load(os.path.join(F, 'checkpoint_set.npz'))
    it = int(cps['it'])
    for sub in meta['run_dir_layout']:
        os.makedirs(os.path.join(str(tmp_path), sub))
    cp = os.path.join(str(tmp_path), 'checkpoint')
    for k in ('active_u', 'active_v', 'active_logl', 'active_derived'):
        np.save(os.path.join(cp, '%s_%d.npy' % (k, it)), cps[k])
    for k in ('saved_v', 'saved_logl', 'saved_logwt'):
        np.save(os.path.join(cp, '%s.npy' % k), cps[k])
    with open(os.path.join(cp, 'checkpoint_%d.txt' % it), 'w') as f:
        json.dump(meta['checkpoint_state'], f)
    np.random.seed(0)
    torch.manual_seed(0)
    s = _sampler(tmp_path, append_run_num=False)
    assert not s.logs['created']
    s.run(train_iters=100, mcmc_num_chains=10)
    assert s.niter > it and s.ncall > meta['checkpoint_state']['ncall']
    assert len(s.loglikes) == s.niter - 1 + 100 and np.array_equal(s.loglikes[:it], cps['saved_logl'])
    # 11 more iterations of 691 happen here; the evidence is essentially the reference's (-6.0258)
    assert abs(s.logz - meta['final_logz']) < 0.05, (s.logz, meta['final_logz'])


def test_chain_text_writer_is_savetxt_byte_for_byte(tmp_path):
    """Sampler._save_samples (sampler.py:494-511) writes np.savetxt(fmt='%.5E'); the native formatter behind it
    (nnest_format_rows_e5, nnest_amd/utils.write_rows_e5) has to produce the same bytes: magnitudes across the float64 range,
    negative zero, infinities, NaN, values that round up a decade, a header line."""
    from nnest_amd.utils import write_rows_e5
    rng = np.random.RandomState(3)
    a = rng.standard_normal((6000, 7)) * np.exp(rng.uniform(-300, 300, size=(6000, 7)))
    a[5, 3] = np.nan; a[6, 2] = np.inf; a[7, 1] = -np.inf; a[8, 0] = 0.0; a[9, 0] = -0.0; a[10, 0] = 1e-300
    a[11, 1] = 9.999995e5; a[12, 2] = 9.9999949e5; a[13, 3] = 1e-30; a[14, 4] = 5e-324
    for header in ('', 'weight minusloglike a b c d e'):
        write_rows_e5(str(tmp_path / 'n.txt'), a, header=header)
        np.savetxt(str(tmp_path / 's.txt'), a, fmt='%.5E', header=header, comments='#')
        assert open(str(tmp_path / 'n.txt'), 'rb').read() == open(str(tmp_path / 's.txt'), 'rb').read()


def test_growing_npy_is_a_valid_npy_after_every_sync(tmp_path):
    """the checkpoint's saved_v / saved_logl / saved_logwt (nested.py:479-481) grow by appended rows; np.load -- what the
    reference's resume does (nested.py:191-193) -- reads them at any point"""
    from nnest_amd.utils import GrowingNpy
    g = GrowingNpy(str(tmp_path / 'v.npy'), (3,))
    s = GrowingNpy(str(tmp_path / 'l.npy'), ())
    rows, vals = [], []
    assert np.load(str(tmp_path / 'v.npy')).shape == (0, 3) and np.load(str(tmp_path / 'l.npy')).shape == (0,)
    for k in range(1, 40):
        rows += [np.arange(3) + 10.0 * k + j for j in range(k % 5)]
        vals += [float(k)] * (k % 3)
        g.sync(rows)
        s.sync(vals)
        assert np.array_equal(np.load(str(tmp_path / 'v.npy')), np.array(rows).reshape(-1, 3))
        assert np.array_equal(np.load(str(tmp_path / 'l.npy')), np.array(vals))


def test_scalar_rows_are_pythons_repr_byte_for_byte():
    """scalars.csv (the stand-in for trainer.writer.add_scalar, nnest/nested.py:467): the bulk rows are formatted by the native library
    (nnest_format_scalar_rows, no interpreter lock held on the run's worker thread) and must be the text the single-row path writes
    -- '%s,%s,%r' -- for every double: shortest round-trip digits, '.0' after integral values, exponent form below 1e-4 and from 1e16,
    signed zero, denormals, nan and the infinities."""
    from nnest_amd.utils import _format_scalar_rows
    rng = np.random.default_rng(5)
    v = np.concatenate([rng.standard_normal(40000) * 10.0 ** rng.integers(-30, 30, 40000),
                        rng.integers(-10 ** 17, 10 ** 17, 5000).astype(float),
                        np.frombuffer(rng.bytes(8 * 20000), dtype=np.float64),
                        [-2.0, 0.75, 0.0, -0.0, 1e-5, 1.5e-5, 1e-4, 9.9e-5, 123456789012345.0, 1e15, 9999999999999998.0, 1e16, 1e17, 0.1,
                         1 / 3, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, np.inf, -np.inf, np.nan, -241.37396826186236]])
    k = rng.integers(-2 ** 62, 2 ** 62, len(v))
    got = _format_scalar_rows('logz', k, v)
    assert got is not None, 'the native library is part of the build'
    assert got.decode() == ''.join('%s,%s,%r\n' % ('logz', a, b) for a, b in zip(k.tolist(), v.tolist()))
    assert _format_scalar_rows('loss', [], []) == b''
