"""Host plumbing shared by the sampler front-ends: logger factory and the numbered run-directory tree
(same layout as the reference's nnest/utils/logger.py:9-22, :38-75 so its analysis scripts find the files)."""
import numpy as np
import errno
import logging
import os
import sys


def create_logger(name, level=logging.INFO):
    logger = logging.getLogger(name)
    logger.setLevel(level)
    if not logger.handlers:
        handler = logging.StreamHandler(sys.stdout)
        handler.setFormatter(logging.Formatter('[%(name)s] [%(levelname)s] %(message)s'))
        logger.addHandler(handler)
        logger.propagate = False
    return logger


def _mkdir(path):
    try:
        os.makedirs(path)
    except OSError as e:
        if e.errno != errno.EEXIST:
            raise


def get_or_create_run_dir(run_dir, append_run_num=True):
    """<log_dir>/run<N>/{info,results,chains,checkpoint,plots}; an existing tree (has info/) is reused."""
    if os.path.isdir(os.path.join(run_dir, 'info')):
        created = False
    else:
        created = True
        _mkdir(run_dir)
        if append_run_num:
            run_num = sum(os.path.isdir(os.path.join(run_dir, i)) for i in os.listdir(run_dir)) + 1
            run_dir = os.path.join(run_dir, 'run%s' % run_num)
        for sub in ('', 'info', 'results', 'chains', 'checkpoint', 'plots'):
            _mkdir(os.path.join(run_dir, sub))
    return {'run_dir': run_dir, 'info': os.path.join(run_dir, 'info'), 'results': os.path.join(run_dir, 'results'),
            'chains': os.path.join(run_dir, 'chains'), 'checkpoint': os.path.join(run_dir, 'checkpoint'),
            'plots': os.path.join(run_dir, 'plots'), 'created': created}


class BackgroundJobs(object):
    """One worker thread that runs queued callables in order (file writes the run does not have to wait for: models/netG.pt after
    every retrain, the bulk log-Z scalars).  A thread per job costs ~0.9 ms in Thread.start() alone -- 0.35 s over the 392 retrains
    of a config-2 run; a queue hand-over is microseconds.  wait(): everything queued so far is done; the first exception a job
    raised is re-raised there."""

    def __init__(self):
        import queue
        self._q = queue.Queue()
        self._thread = None
        self._err = None

    def _loop(self):
        while True:
            fn = self._q.get()
            try:
                fn()
            except BaseException as e:   # kept for wait()
                if self._err is None:
                    self._err = e
            finally:
                self._q.task_done()

    def submit(self, fn):
        import threading
        if self._thread is None or not self._thread.is_alive():
            self._thread = threading.Thread(target=self._loop, daemon=True)   # (daemon: owners call wait() before they rely on the files)
            self._thread.start()
        self._q.put(fn)

    def wait(self):
        if self._thread is not None:
            self._q.join()
        if self._err is not None:
            e, self._err = self._err, None
            raise e


def _format_scalar_rows(tag, steps, values):
    """the rows '<tag>,<step>,<repr(value)>\\n' of a run of steps as bytes, formatted by the native library
    (nnest_format_scalar_rows: no interpreter lock held, which matters on the worker thread of a run -- 230 000 rows at config 2);
    None where the library is not at hand"""
    try:
        import ctypes
        from . import _lib
        lib = _lib.load()
        ks = np.ascontiguousarray(steps, dtype=np.int64)
        vs = np.ascontiguousarray(values, dtype=np.float64)
        if ks.ndim != 1 or ks.shape != vs.shape:
            return None
        tagb = str(tag).encode()
        cap = ks.size * (len(tagb) + 48) + 1
        buf = np.empty(cap, dtype=np.uint8)
        n = lib.nnest_format_scalar_rows(tagb, ks.ctypes.data_as(ctypes.c_void_p), vs.ctypes.data_as(ctypes.c_void_p), ks.size,
                                         buf.ctypes.data_as(ctypes.c_void_p), cap)
        return None if n < 0 else buf[:n].tobytes()
    except (OSError, AttributeError, ValueError, RuntimeError, TypeError):
        return None


class ScalarWriter(object):
    """Stands where the reference keeps a TensorBoard SummaryWriter (trainer.py:127-129; used by
    nested.py:467): scalars are appended to <path>/scalars.csv; figures are dropped."""

    def __init__(self, path=None):
        self.path = None if path is None else os.path.join(path, 'scalars.csv')
        self._buf = []
        self.jobs = None    # a BackgroundJobs: bulk rows are formatted and appended there (NestedSampler.run lends the trainer's)

    # Rows are kept as they arrive -- single rows as text, bulk rows as (tag, steps, values) arrays -- and turned into text when
    # there is something to gain from writing: every WRITE_EVERY seconds on the worker thread where there is one (a config-2 run
    # logs 230 000 rows; formatting them as they came held the interpreter lock for a quarter of a second of the run), at flush()
    # at the latest.  The file keeps the order of the calls.
    WRITE_EVERY = 0.5

    def add_scalar(self, tag, value, step=None):
        if self.path is None:
            return
        self._buf.append('%s,%s,%r\n' % (tag, step, float(value)))
        self._maybe_write()

    def add_scalars(self, tag, steps, values):
        """add_scalar for a run of steps at once (the native nested-sampling loop reports log Z per accepted point in bulk)"""
        if self.path is None:
            return
        self._buf.append((tag, np.array(steps), np.array(values, dtype=np.float64)))
        self._maybe_write()

    def _maybe_write(self):
        import time
        now = time.time()
        if self.jobs is None:
            if len(self._buf) >= 256:
                self._write(self._take())
        elif now - getattr(self, '_last_write', 0.0) >= self.WRITE_EVERY:
            self._last_write = now
            rows = self._take()
            self.jobs.submit(lambda: self._write(rows))

    def _take(self):
        rows, self._buf = self._buf, []
        return rows

    def _write(self, rows):
        out = []
        for r in rows:
            if isinstance(r, str):
                out.append(r.encode())
            else:
                tag, ks, vs = r
                text = _format_scalar_rows(tag, ks, vs)
                if text is None:
                    text = ''.join(['%s,%s,%r\n' % (tag, k, v) for k, v in zip(ks.tolist(), vs.tolist())]).encode()
                out.append(text)
        if out:
            with open(self.path, 'ab') as f:
                f.writelines(out)

    def add_figure(self, *a, **k):
        pass

    def flush(self):
        if self.jobs is not None:
            self.jobs.wait()
        if self.path is not None and self._buf:
            self._write(self._take())

    def __del__(self):
        try:
            self.flush()
        except Exception:
            pass


class GrowingNpy(object):
    """An .npy file (format 1.0, float64, C order) that grows by appended rows: the header is written once with room for the
    row count and patched in place, so that a dump costs the new rows only -- the reference rewrites every dead point at every
    checkpoint (nested.py:473-485: np.save of the whole list), which is quadratic over a run.  np.load reads the file like
    any other .npy at any time between two `sync` calls."""
    HEADER_BYTES = 128

    def __init__(self, path, row_shape, initial=None):
        """initial: rows the file starts with (a run resumed from a checkpoint: the dead points read back) -- written to a
        temporary file that then REPLACES `path`, so the rows on disk are never gone, whoever wrote `path` before"""
        self.path, self.row_shape = path, tuple(int(v) for v in row_shape)
        self._rewrite(initial)

    def _header(self, n):
        d = "{'descr': '<f8', 'fortran_order': False, 'shape': %s, }" % (repr((n,) + self.row_shape),)
        body = self.HEADER_BYTES - 10
        assert len(d) < body
        return b'\x93NUMPY\x01\x00' + np.uint16(body).tobytes() + (d + ' ' * (body - 1 - len(d)) + '\n').encode('latin1')

    def _rewrite(self, rows):
        n = 0 if rows is None else len(rows)
        tmp = self.path + '.tmp'
        with open(tmp, 'wb') as f:
            f.write(self._header(n))
            if n:
                f.write(np.ascontiguousarray(np.asarray(rows, dtype=np.float64)).reshape((n,) + self.row_shape).tobytes())
        os.replace(tmp, self.path)
        self.rows = n

    def sync(self, rows):
        """`rows`: the complete sequence so far (list of rows or array); rows beyond those already on disk are appended.  A
        sequence SHORTER than the file (a new run on the same object) rewrites the file."""
        n = len(rows)
        if n < self.rows:
            self._rewrite(rows)
        elif n > self.rows:
            new = np.ascontiguousarray(np.asarray(rows[self.rows:n], dtype=np.float64)).reshape((n - self.rows,) + self.row_shape)
            with open(self.path, 'r+b') as f:
                f.seek(0, 2)
                f.write(new.tobytes())
                f.seek(0)
                f.write(self._header(n))
            self.rows = n


def write_rows_e5(path, rows, header=''):
    """np.savetxt(path, rows, fmt='%.5E', header=header, comments='#') -- the chain-file format of Sampler._save_samples
    (sampler.py:494-511) -- with the rows formatted by the native library (nnest_format_rows_e5: snprintf on a few host threads;
    np.savetxt formats row by row in Python, 1.8 s for a 2e5 x 52 chain)."""
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    if rows.ndim == 1:
        rows = rows[:, None]
    try:
        from . import _lib
        import ctypes
        lib = _lib.load()
        cap = 14 * rows.size + 1
        buf = np.empty(cap, dtype=np.uint8)    # (not zero-filled: 1.4 GB for a config-5 chain of 1e6 x 102)
        threads = min(64, max(1, os.cpu_count() or 1))
        n = lib.nnest_format_rows_e5(rows.ctypes.data_as(ctypes.c_void_p), rows.shape[0], rows.shape[1],
                                     buf.ctypes.data_as(ctypes.c_void_p), cap, threads)
        if n < 0:
            raise ValueError('nnest_format_rows_e5')
        with open(path, 'wb') as f:
            if header:
                f.write(('#' + header + '\n').encode())
            f.write(memoryview(buf)[:n])
    except (OSError, AttributeError, ValueError, RuntimeError):   # no native library at hand (a pure-host use of the driver)
        np.savetxt(path, rows, fmt='%.5E', header=header, comments='#')
