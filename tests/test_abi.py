"""CPU checks of the C-ABI boundary: the library loads and exports every symbol include/nnest_hip.h declares
(no compute calls without a GPU), and the product refuses to run without a GPU instead of falling back."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'nnest_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(nnest_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_expected_surface():
    syms = declared_symbols()
    for must in ('nnest_nvp_forward', 'nnest_nvp_inverse', 'nnest_nvp_log_probs', 'nnest_nvp_inverse_loglike',
                 'nnest_loglike', 'nnest_mh_constrained_steps', 'nnest_nvp_train', 'nnest_hip_version'):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from nnest_amd import _lib
    lib = _lib.load()
    syms = declared_symbols()
    assert set(syms) == set(_lib.SIGNATURES), set(syms) ^ set(_lib.SIGNATURES)
    for s in syms:
        assert hasattr(lib, s), s
    assert lib.nnest_hip_version() == 15


def test_argument_errors_are_reported_not_thrown():
    from nnest_amd import _lib
    lib = _lib.load()
    import ctypes
    h = ctypes.c_void_p()
    rc = lib.nnest_nvp_create(50, 17, 3, 1, ctypes.byref(h))  # hidden_dim not a multiple of 16
    assert rc == 3 and b'hidden_dim' in lib.nnest_hip_last_error()
    rc = lib.nnest_nvp_create(0, 16, 3, 1, ctypes.byref(h))
    assert rc == 1
    assert lib.nnest_nvp_forward(None, None, None, None, 4, None) == 1


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from nnest_amd import _lib
    from nnest_amd.flow import HipNVP
    from nnest_amd.trainer import Trainer
    with pytest.raises(_lib.NnestHipError):
        HipNVP(4)
    with pytest.raises(_lib.NnestHipError):
        Trainer(4, log_dir=None)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'nnest_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in text.lower(), (f, 'the product must not reference the oracle')
