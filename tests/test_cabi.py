"""CPU: the C-ABI shared library loads and exports every symbol that include/vargp_hip.h declares;
the product refuses CPU tensors (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, 'include', 'vargp_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(vargp_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from vargp_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), 'build with make -C vargp_amd/csrc'
    handle = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(handle, n), f'{n} declared in include/vargp_hip.h but not exported'
    # and the Python binding covers the same set
    assert sorted(_lib.EXPORTS) == names
    assert _lib.lib().vargp_version() >= 100


def test_no_cpu_fallback():
    from vargp_amd import ops
    from vargp_amd._lib import VargpHipError
    a = torch.randn(2, 4, 4)
    with pytest.raises(VargpHipError):
        ops.matmul(a, a)
    with pytest.raises(VargpHipError):
        ops.chol(a @ a.mT)
    with pytest.raises(VargpHipError):
        ops.rbf_gram(torch.zeros(1, 3), torch.zeros(1, 4, 2))


def test_product_does_not_import_oracle():
    """vargp_amd/ must never reach into oracle/ (the checker is not the product)."""
    pkg = os.path.join(ROOT, 'vargp_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), f
