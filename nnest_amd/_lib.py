"""ctypes binding of libnnest_hip.so (include/nnest_hip.h).

The shared library is the product: there is no CPU or PyTorch fallback.  Importing this module
without the built library raises; calling into it without a GPU returns the library's own error.
`import torch` happens first so that the library's libamdhip64.so.7 dependency resolves to the HIP
runtime PyTorch-ROCm has already loaded (one runtime per process; device pointers and streams are
shared with torch tensors).
"""
import os
import ctypes

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('NNEST_HIP_LIB', os.path.join(_HERE, 'libnnest_hip.so'))  # override: developer A/B builds only

NNEST_OK = 0
NNEST_E_UNSUPPORTED = 3
LIKE_IDS = {'rosenbrock': 0, 'gaussmix': 1, 'himmelblau': 2, 'gaussian': 3, 'eggbox': 4, 'shell': 5, 'double_shell': 6}
MH_DYNAMIC_STEP = 1      # per 16-walker group
MH_UNCONSTRAINED = 2
MH_DYNAMIC_BATCH = 4     # over the whole launch, as the reference (sampler.py:422-431); lag in bits 8..11
MH_FORMS = {None: 0, 'auto': 0, 'image': 1, 'reg': 2, 'team': 3, 'quad': 4, 'quad1': 5, 'solo': 6}
MH_FORM_NAMES = {v: k for k, v in MH_FORMS.items() if isinstance(k, str) and v}
MH_DEFAULT_LAG = 4      # steps between a step and the scale that reflects its batch-wide count
MH_WARM_STEPS = 16      # exact steps in front of the lagged rule where the form implements them (NNEST_MH_WARM)
MH_SYNC_ZERO_NEXT, MH_SYNC_ZERO_PREV = 1 << 29, 1 << 30   # sync_dev as one half of a double buffer (include/nnest_hip.h)
MH_ALL_MOVED = 1 << 30  # n_accept words: every coordinate of the chain's last x differs from its first (include/nnest_hip.h)
MH_SOLO_LAG = 8         # ... where the solo form runs (its steps are shorter: the same ~10 us of latency)
TRAIN_RESUME = 1
TRAIN_FINALIZE = 2
TRAIN_ONE_CU = 4


def mh_flags(dynamic=False, free=False, lag=None, form=None, warm=0):
    """flags word of nnest_mh_constrained_steps.  dynamic: False | True / 'batch' (the reference's batch-wide rule) |
    'group' (per 16 walkers); warm: NNEST_MH_WARM, exact steps in front of a lagged batch rule"""
    fl = MH_UNCONSTRAINED if free else 0
    if dynamic == 'group':
        fl |= MH_DYNAMIC_STEP
    elif dynamic:
        fl |= MH_DYNAMIC_BATCH | ((MH_DEFAULT_LAG if lag is None else int(lag)) & 15) << 8
        fl |= (min(int(warm or 0), 255) & 255) << 20
    return fl | (MH_FORMS[form] << 16)


class NnestHipError(RuntimeError):
    code = None   # NNEST_E_* of the failing call, when it came from the library


class LikeSpec(ctypes.Structure):
    """nnest_like_t (include/nnest_hip.h): likelihood id, transform scale, parameters"""
    _fields_ = [('id', ctypes.c_int), ('scale', ctypes.c_float), ('params', ctypes.c_float * 6)]


def like_spec(like_id, scale, params=None):
    lk = LikeSpec()
    lk.id = int(like_id)
    lk.scale = float(scale)
    for i, v in enumerate(params or ()):
        lk.params[i] = float(v)
    return lk


class TrainResult(ctypes.Structure):
    _fields_ = [('epochs_run', ctypes.c_int), ('best_epoch', ctypes.c_int),
                ('best_validation_loss', ctypes.c_float), ('last_train_loss', ctypes.c_float),
                ('counter', ctypes.c_int), ('stopped', ctypes.c_int)]


_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_d = ctypes.c_double
_u64 = ctypes.c_uint64

# name -> argtypes; every entry point include/nnest_hip.h declares (checked by tests/test_abi.py)
SIGNATURES = {
    'nnest_hip_version': [],
    'nnest_hip_last_error': [],
    'nnest_hip_device_info': [ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.c_char_p, _i],
    'nnest_nvp_create': [_i, _i, _i, _i, ctypes.POINTER(_vp)],
    'nnest_spline_create': [_i, _i, _i, _i, _f, ctypes.POINTER(_vp)],
    'nnest_spline_destroy': [_vp],
    'nnest_spline_num_params': [_vp],
    'nnest_spline_load_weights': [_vp, _vp, _vp, _vp],
    'nnest_spline_store_weights': [_vp, _vp, _vp, _vp],
    'nnest_nvp_create_scaled': [_i, _i, _i, _i, _i, ctypes.POINTER(_vp)],
    'nnest_maf_create': [_i, _i, _i, _i, ctypes.POINTER(_vp)],
    'nnest_maf_num_groups': [_vp],
    'nnest_maf_train_epoch': [_vp, _vp, _i, _i, _f, _f, _vp, _vp],
    'nnest_nvp_destroy': [_vp],
    'nnest_nvp_num_params': [_vp],
    'nnest_nvp_set_base': [_vp, _f],
    'nnest_nvp_vjp': [_vp, _vp, _vp, _f, _i, _vp, _vp, _vp],
    'nnest_nvp_adam_step': [_vp, _vp, _f, _f, _vp],
    'nnest_spline_set_base': [_vp, _f],
    'nnest_nvp_load_weights': [_vp, _vp, _vp],
    'nnest_nvp_store_weights': [_vp, _vp, _vp],
    'nnest_nvp_device_ptrs': [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp)],
    'nnest_nvp_store_adam': [_vp, _vp, _vp, _vp],
    'nnest_nvp_load_adam': [_vp, _vp, _vp, _vp],
    'nnest_nvp_adam_state': [_vp, ctypes.POINTER(_i), _i, _i, _vp],
    'nnest_nvp_forward': [_vp, _vp, _vp, _vp, _i, _vp],
    'nnest_nvp_inverse': [_vp, _vp, _vp, _vp, _i, _vp],
    'nnest_nvp_log_probs': [_vp, _vp, _vp, _i, _vp],
    'nnest_nvp_inverse_loglike': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    'nnest_loglike': [_vp, _vp, _vp, _i, _i, _vp],
    'nnest_mh_constrained_steps': [_vp, _vp, _vp, _vp, _vp, _d, _f, _i, _i, _i, _vp, _vp, _u64, _u64,
                                   _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    'nnest_mh_sync_words': [_i],
    'nnest_mh_form_for': [_vp, _i, _i],
    'nnest_spline_mh_form_for': [_vp, _i, _i],
    'nnest_spline_train_form': [_vp, _i],
    'nnest_spline_forward': [_vp, _vp, _vp, _vp, _i, _vp],
    'nnest_spline_inverse': [_vp, _vp, _vp, _vp, _i, _vp],
    'nnest_spline_log_probs': [_vp, _vp, _vp, _i, _vp],
    'nnest_spline_inverse_loglike': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    'nnest_spline_mh_constrained_steps': [_vp, _vp, _vp, _vp, _vp, _d, _f, _i, _i, _i, _vp, _vp, _u64, _u64,
                                          _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    'nnest_spline_vjp': [_vp, _vp, _vp, _f, _i, _vp, _vp, _vp],
    'nnest_spline_adam_step': [_vp, _vp, _f, _f, _vp],
    'nnest_spline_actnorm_init': [_vp, _vp, _i, _vp],
    'nnest_spline_loss_grad': [_vp, _vp, _i, _vp, _vp, _vp],
    'nnest_spline_train': [_vp, _vp, _i, _vp, _i, _vp, _vp, _u64, _f, _i, _i, _i, _f, _f, _vp, _vp, _vp],
    'nnest_chol_create': [_i, ctypes.POINTER(_vp)],
    'nnest_chol_destroy': [_vp],
    'nnest_chol_num_params': [_vp],
    'nnest_chol_set_base': [_vp, _f],
    'nnest_chol_load_weights': [_vp, _vp, _vp],
    'nnest_chol_store_weights': [_vp, _vp, _vp],
    'nnest_chol_forward': [_vp, _vp, _vp, _vp, _i, _vp],
    'nnest_chol_inverse': [_vp, _vp, _vp, _vp, _i, _vp],
    'nnest_chol_log_probs': [_vp, _vp, _vp, _i, _vp],
    'nnest_chol_loss_grad': [_vp, _vp, _i, _vp, _vp, _vp],
    'nnest_chol_adam_step': [_vp, _vp, _f, _f, _vp],
    'nnest_mh_num_groups': [_vp, _i],
    'nnest_mh_fill_noise': [_vp, _vp, _i, _i, _i, _u64, _u64, _vp],
    'nnest_nvp_train': [_vp, _vp, _i, _vp, _i, _vp, _vp, _u64, _f, _i, _i, _i, _f, _f, _i, _i, _vp, _vp, _vp],
    'nnest_nvp_loss_grad': [_vp, _vp, _i, _vp, _vp, _vp],
    'nnest_training_jitter': [_vp, _i, _i, _vp, _vp],
    'nnest_format_rows_e5': [_vp, ctypes.c_long, _i, _vp, ctypes.c_long, _i],
    'nnest_format_scalar_rows': [ctypes.c_char_p, _vp, _vp, ctypes.c_long, _vp, ctypes.c_long],
    'nnest_host_mcmc_consume': [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, ctypes.c_longlong,
                                _d, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong],
    'nnest_host_h_update': [_d, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong],
    'nnest_slice_steps': [_vp, _vp, _vp, _vp, _vp, _d, _f, _i, _i, _i, _i, _vp, _u64, _u64, _vp, _vp, _vp, _vp, _vp],
    'nnest_slice_fill_noise': [_vp, _i, _i, _i, _u64, _u64, _vp],
    'nnest_host_prior_consume': [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                 ctypes.c_longlong, _d, ctypes.c_longlong, ctypes.c_longlong, _d, _d, _i],
}
HOST_FINISHED, HOST_RETRAIN, HOST_NEED_SAMPLES, HOST_LOG, HOST_CHECKPOINT, HOST_DEAD_FULL = range(6)   # include/nnest_hip.h NNEST_HOST_*
HOST_EXPIRED = 6
HOST_TOP, HOST_AFTER_TRAIN, HOST_AFTER_SAMPLES, HOST_AFTER_LOG = range(4)


class HostState(ctypes.Structure):   # nnest_host_state_t
    _fields_ = [('logz', _d), ('logvol', _d), ('fraction_remain', _d), ('max_logl', _d), ('loglstar', _d),
                ('it', ctypes.c_longlong), ('n_dead', ctypes.c_longlong),
                ('accept_point', _i), ('nb', _i), ('first_time', _i), ('resume', _i), ('worst', _i), ('pad_', _i)]


class HostPrior(ctypes.Structure):   # nnest_host_prior_t
    _fields_ = [('pos', ctypes.c_longlong), ('k', ctypes.c_longlong), ('hits', ctypes.c_longlong), ('n', ctypes.c_longlong),
                ('n_cand', ctypes.c_longlong), ('pending_calls', ctypes.c_longlong), ('total_calls', ctypes.c_longlong),
                ('block_next', ctypes.c_longlong), ('ncs', _d * 20), ('mean_calls', _d), ('ncs_len', _i), ('expired', _i)]


_lib = None


def load():
    """Load libnnest_hip.so (built by __graft_entry__.build() / make -C nnest_amd/csrc)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NnestHipError('%s is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                                '(hipcc --offload-arch=gfx950).  There is no CPU fallback.' % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
            fn.argtypes = argtypes
            fn.restype = (ctypes.c_char_p if name == 'nnest_hip_last_error' else ctypes.c_long if name in ('nnest_format_rows_e5', 'nnest_format_scalar_rows')
                          else ctypes.c_double if name == 'nnest_host_h_update' else ctypes.c_int)
        _lib = lib
    return _lib


def check(rc):
    if rc != NNEST_OK:
        msg = load().nnest_hip_last_error()
        err = NnestHipError('libnnest_hip error %d: %s' % (rc, msg.decode() if msg else '?'))
        err.code = rc
        raise err


def ptr(t):
    """device pointer of a torch tensor (or None)"""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def current_stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def device_info():
    n = ctypes.c_int(0)
    clk = ctypes.c_int(0)
    buf = ctypes.create_string_buffer(256)
    check(load().nnest_hip_device_info(ctypes.byref(n), ctypes.byref(clk), buf, 256))
    return {'num_cu': n.value, 'clock_khz': clk.value, 'name': buf.value.decode()}
